"""Host-side checks that need no GPU: the C-ABI library exports every symbol the header declares, the ctypes mirror of
crog_gemm_desc matches the C struct, argument validation works without touching a device, and the module tree reproduces
the reference's parameter names / shapes / optimizer groups (fixtures captured from the reference, tests/golden/*.json)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd import _lib  # noqa: E402
from crog_amd.testing import make_cfg, tiny_cfg  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    protos = _lib.parse_header()
    assert len(protos) >= 45
    for name in protos:
        assert hasattr(lib, name), name
    # and the header is the only place prototypes live: every extern "C" crog_* definition is declared there
    defined = set()
    for f in _lib.SOURCES:
        defined |= set(re.findall(r'extern "C" (?:const char\*|int) (crog_\w+)\(', open(os.path.join(_lib.CSRC, f)).read()))
    assert defined == set(protos), (defined ^ set(protos))
    assert lib.crog_hip_version() >= 100


def test_gemm_desc_layout_matches_c_struct(tmp_path):
    """Every field of crog_gemm_desc: same order, same offset, same total size in the C header and in the ctypes mirror."""
    D = _lib.GemmDesc
    names = [f[0] for f in D._fields_]
    header = open(os.path.join(ROOT, "include", "crog_hip.h")).read()
    body = re.search(r"typedef struct crog_gemm_desc \{(.*?)\} crog_gemm_desc;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    c_names = []
    for decl in body.split(";"):          # "int M, N, K" declares three fields
        if decl.strip():
            c_names += [re.split(r"[\s\*]+", part.strip())[-1] for part in decl.split(",")]
    assert c_names == names, (c_names, names)
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "crog_hip.h"\nint main(){printf("%zu", sizeof(crog_gemm_desc));'
                   + "".join('printf(" %%zu", offsetof(crog_gemm_desc, %s));' % n for n in names) + 'printf("\\n");return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out == [ctypes.sizeof(D)] + [getattr(D, n).offset for n in names]


def test_argument_validation_needs_no_device(lib):
    d = _lib.GemmDesc()
    assert lib.crog_gemm(ctypes.byref(d), None) == 0           # empty problem: nothing to do
    d.M, d.N, d.K, d.batch = 4, 4, 4, 1
    assert lib.crog_gemm(ctypes.byref(d), None) == -1          # null operands are rejected before any launch
    assert b"crog_gemm" in lib.crog_last_error()
    d.dtype = 7
    assert lib.crog_gemm(ctypes.byref(d), None) == -1 and b"dtype" in lib.crog_last_error()
    # BatchNorm-backward statistics mode: needs the atomic replica form of col_stats (checked before any launch)
    buf = (ctypes.c_char * 4096)()
    addr = (ctypes.addressof(buf) + 15) // 16 * 16
    e = _lib.GemmDesc(dtype=1, a_layout=0, b_layout=1, A=addr, B=addr, C=addr, M=8, N=8, K=8, lda=8, ldb=8, ldc=8, batch=1, batch_inner=1,
                      splitk=1, alpha=1.0, bwd_z=addr, ldz=8)
    assert lib.crog_gemm(ctypes.byref(e), None) == -1 and b"bwd_z" in lib.crog_last_error()
    t = ctypes.c_void_p()
    assert lib.crog_timer_record(None, None) == -1 and lib.crog_timer_elapsed_ms(None, None, None) == -1 and lib.crog_timer_create(None) == -1
    assert lib.crog_gemm_stat_tiles(129) == 2
    assert lib.crog_adam_step(None, None, None, None, 0, 0.0, 0.9, 0.999, 1e-8, 0.0, 0, None, None) == -1  # step must be >= 1


def test_product_path_has_no_cpu_fallback():
    from crog_amd import kernels as K
    with pytest.raises(RuntimeError):
        K.ptr(torch.zeros(4))                                    # CPU tensors are refused, not silently computed
    from crog_amd.model import build_crog
    model, _ = build_crog(tiny_cfg())
    with pytest.raises(RuntimeError):
        model.prepare(torch.device("cpu"))
    # nothing under crog_amd/ imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "crog_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


@pytest.mark.parametrize("case,cfgf", [("tiny_crog", tiny_cfg), ("crog_r50_b2", make_cfg)])
def test_state_dict_and_groups_match_reference(case, cfgf):
    from crog_amd.model import build_crog
    meta = json.load(open(os.path.join(GOLD, case + ".json")))
    model, groups = build_crog(cfgf())
    sd = model.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["shapes"]
    assert len(groups[0]["params"]) == meta["group_backbone"] and len(groups[1]["params"]) == meta["group_head"]
    assert [groups[0]["initial_lr"], groups[1]["initial_lr"]] == pytest.approx(meta["group_lrs"])
    names = [n for n, _ in model.named_parameters()]
    assert sorted(names) == sorted(meta["param_names"])
    if case == "crog_r50_b2":
        assert sum(p.numel() for p in model.parameters()) == 147112290


def test_flat_store_layout_on_cpu_tensors():
    """ParamStore is plain tensor bookkeeping (no kernels): check offsets, alignment, KRSC views and grad aliasing on CPU."""
    from crog_amd.runtime import ALIGN, ParamStore
    m = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, bias=False), torch.nn.Conv2d(16, 4, 1), torch.nn.Linear(5, 3))
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    st = ParamStore(m, torch.device("cpu"))
    for name, p, o, n, g in st.entries:
        assert o % ALIGN == 0 and torch.equal(p.detach(), ref[name]) and p.grad is g
        assert p.data_ptr() == st.P.data_ptr() + 4 * o
    w = m[0].weight
    assert w.shape == (16, 8, 3, 3) and w.stride() == (72, 1, 24, 8)          # physically [Cout][ky][kx][Cin]
    assert torch.equal(st.P[:16 * 72].view(16, 3, 3, 8), ref["0.weight"].permute(0, 2, 3, 1))
    out = m[2](m[1](m[0](torch.randn(2, 8, 6, 6))).flatten(1)[:, :5])
    out.sum().backward()
    assert st.G.abs().sum() > 0 and m[0].weight.grad.data_ptr() == st.G.data_ptr()   # autograd accumulated in place into G
    st.zero_grad()
    assert m[2].weight.grad.abs().sum() == 0


def test_checkpoint_prefix_handling(tmp_path):
    """train_crog.py:213: the reference loads `module.`-prefixed keys into its DDP wrapper; crog_amd.checkpoint accepts either side
    wrapped or bare, and refuses files that are not CROG checkpoints."""
    import torch
    from crog_amd.checkpoint import KEYS, load_checkpoint, save_checkpoint

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m

    bare = torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.BatchNorm1d(2))
    opt = torch.optim.Adam(bare.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    p = str(tmp_path / "a.pth")
    ck = save_checkpoint(p, Wrap(bare), opt, sched, epoch=4, best_iou=0.7)
    assert tuple(ck.keys()) == KEYS and all(k.startswith("module.") for k in ck["state_dict"])
    other = torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.BatchNorm1d(2))
    info = load_checkpoint(p, other, map_location="cpu")                      # wrapped file -> bare model
    assert info["epoch"] == 4 and info["best_iou"] == 0.7 and torch.equal(other[0].weight, bare[0].weight)
    p2 = str(tmp_path / "b.pth")
    save_checkpoint(p2, bare, opt, sched, epoch=1)
    load_checkpoint(p2, Wrap(other), map_location="cpu")                      # bare file -> wrapped model
    torch.save({"weights": 1}, p)
    import pytest
    with pytest.raises(KeyError):
        load_checkpoint(p, other)


def test_device_code_is_built_without_packed_fp32_instructions(tmp_path):
    """crog_amd/_lib.py NO_PACKED_F32 (LAB_NOTES section 10): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 return wrong results in lanes 48-63
    while another stream's MFMA kernel shares the SIMD, so no kernel of the library may contain them.  Compiles the source that held the kernel
    the effect was found in (eltwise.hip: 728 of them under plain -O3) to ISA with the library's flags and looks."""
    assert _lib.NO_PACKED_F32 and all(f in _lib.HIPCC_FLAGS for f in _lib.NO_PACKED_F32)
    out = tmp_path / "eltwise.s"
    flags = [f for f in _lib.HIPCC_FLAGS if f != "-fPIC"]
    subprocess.check_call(["hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(_lib.CSRC, "eltwise.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    assert "upsample2_bwd_quad_kernel" in text
    assert not re.search(r"\bv_pk_(fma|mul|add)_f32\b", text)


def test_attention_backward_loops_hold_no_per_score_branches(tmp_path):
    """Round 5 (LAB_NOTES section 10): hipcc had compiled the backward attention kernels' `cond ? exp2(x) : 0` per score into an exec-mask
    branch per score - in dK/dV 16 per tile, each with a ds_read_b32 and its own wait inside (44 s_and_saveexec in the loop, 231 us per decoder
    layer).  Written as the exponential of a selected argument over bitwise conditions the loop is straight-line code (17 / 5 / 1 branches left:
    prefetch guards and uniform paths; 180 us).  Compiles attn.hip to ISA with the library's flags and counts inside each kernel's main loop."""
    out = tmp_path / "attn.s"
    flags = [f for f in _lib.HIPCC_FLAGS if f != "-fPIC"] + _lib.EXTRA_FLAGS.get("attn.hip", [])
    subprocess.check_call(["hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(_lib.CSRC, "attn.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for line in out.read_text().splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur:
            kernels[cur].append(line)
    limits = {"flash_fwd_kernel": 6, "flash_bwd_dq_kernel": 12, "flash_bwd_dkdv_kernel": 24, "flash_bwd_dkdv_short_kernel": 14}
    seen = set()
    for name, lines in kernels.items():
        key = next((k for k in limits if re.search(rf"\d+{k}E", name)), None)
        if key is None:
            continue
        seen.add(key)
        hdr = [i for i, l in enumerate(lines) if "Loop Header" in l]
        end = max(i for i, l in enumerate(lines) if "in Loop: Header" in l or "Loop Header" in l)
        body = lines[hdr[0]:end + 40]
        n = sum("s_and_saveexec" in l for l in body)
        assert sum("v_mfma" in l for l in body) >= 8, name
        assert n <= limits[key], f"{key}: {n} exec-mask branches inside the main loop (limit {limits[key]})"
        assert not any("ds_read_b32 " in l for l in body if key == "flash_bwd_dkdv_kernel"), "lse / D must be read as 16-byte vectors"
    assert seen == set(limits), seen
