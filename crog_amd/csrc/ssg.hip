// SSG target assignment on the device: anchor <-> ground-truth matching of the whole batch in two launches,
// replacing the per-image Python loop of model/ssg.py:317-321 and the per-box loop of utils/box_utils.py:57-85.
//
//   pass 1  (one block per (ground-truth box, image)):   claim[b][g] = argmax_a IoU(gt[b][g], anchor a)      ("each box keeps its best anchor")
//   pass 2  (one thread per (anchor, image)):            best box of the anchor, then the claims in box order (a later box wins a
//            contested anchor, as the reference's sequential loop), label thresholds, SSD offset encoding
//
// HBM-bound integer/index work: A x G IoUs per image (18,525 anchors x <= 32 boxes), every anchor read once per pass.  The arithmetic
// repeats the reference's float operations in the same order with contraction off, so that an IoU sitting on a threshold falls on the
// same side; ties take the lowest index (torch.max's rule).
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NT = 256;

struct Box { float x1, y1, x2, y2; };

__device__ inline Box anchor_corners(const float* a) {      // box_utils.py:59: (cx, cy, w, h) -> corners
  const float hw = a[2] / 2.f, hh = a[3] / 2.f;
  return Box{a[0] - hw, a[1] - hh, a[0] + hw, a[1] + hh};
}
__device__ inline float iou(const Box& g, const Box& p) {   // box_utils.py:8-37
  const float lx = fmaxf(g.x1, p.x1), ly = fmaxf(g.y1, p.y1), hx = fminf(g.x2, p.x2), hy = fminf(g.y2, p.y2);
  const float w = fmaxf(hx - lx, 0.f), h = fmaxf(hy - ly, 0.f);
  const float inter = w * h;
  const float area_g = (g.x2 - g.x1) * (g.y2 - g.y1), area_p = (p.x2 - p.x1) * (p.y2 - p.y1);
  return inter / (area_g + area_p - inter);
}

__global__ void __launch_bounds__(NT) ssg_claim_kernel(const float* __restrict__ anchors, int A, const float* __restrict__ gt, const int* __restrict__ ng,
                                                       int Gmax, int* __restrict__ claim) {
  const int g = blockIdx.x, b = blockIdx.y;
  if (g >= ng[b]) {
    if (threadIdx.x == 0) claim[b * Gmax + g] = -1;
    return;
  }
  const float* q = gt + ((long)b * Gmax + g) * 5;
  const Box box{q[0], q[1], q[2], q[3]};
  float best = -1.f;
  int arg = 0x7fffffff;
  for (int a = threadIdx.x; a < A; a += NT) {
    const float v = iou(box, anchor_corners(anchors + 4 * a));
    if (v > best) { best = v; arg = a; }      // ascending a per thread: the first maximum is kept
  }
  __shared__ float sv[NT];
  __shared__ int si[NT];
  sv[threadIdx.x] = best;
  si[threadIdx.x] = arg;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = sv[threadIdx.x + s];
      const int i = si[threadIdx.x + s];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) claim[b * Gmax + g] = si[0];
}

__global__ void __launch_bounds__(NT) ssg_assign_kernel(const float* __restrict__ anchors, int A, const float* __restrict__ gt, const int* __restrict__ ng,
                                                        int Gmax, const int* __restrict__ claim, float pos_thr, float neg_thr,
                                                        float* __restrict__ offsets, int64_t* __restrict__ labels, float* __restrict__ mbox,
                                                        int64_t* __restrict__ midx) {
  const int a = blockIdx.x * NT + threadIdx.x, b = blockIdx.y;
  if (a >= A) return;
  const float* an = anchors + 4 * a;
  const Box p = anchor_corners(an);
  const float* q = gt + (long)b * Gmax * 5;
  const int n = ng[b];
  float best = -1.f;
  int arg = 0;
  for (int g = 0; g < n; g++) {
    const float v = iou(Box{q[5 * g], q[5 * g + 1], q[5 * g + 2], q[5 * g + 3]}, p);
    if (v > best) { best = v; arg = g; }
  }
  for (int g = 0; g < n; g++)
    if (claim[b * Gmax + g] == a) { arg = g; best = 2.f; }      // box_utils.py:67-71: sequential, the later box wins
  const float* m = q + 5 * arg;
  long lab = (long)m[4];
  if (best < pos_thr) lab = -1;
  if (best < neg_thr) lab = 0;
  const long o = (long)b * A + a;
  labels[o] = lab;
  midx[o] = arg;
  mbox[4 * o + 0] = m[0]; mbox[4 * o + 1] = m[1]; mbox[4 * o + 2] = m[2]; mbox[4 * o + 3] = m[3];
  // box_utils.py:106-117, variances (0.1, 0.2)
  offsets[4 * o + 0] = ((m[0] + m[2]) / 2.f - an[0]) / (0.1f * an[2]);
  offsets[4 * o + 1] = ((m[1] + m[3]) / 2.f - an[1]) / (0.1f * an[3]);
  offsets[4 * o + 2] = logf((m[2] - m[0]) / an[2]) / 0.2f;
  offsets[4 * o + 3] = logf((m[3] - m[1]) / an[3]) / 0.2f;
}

}  // namespace

extern "C" int crog_ssg_match(const float* anchors, int A, const float* gt, const int* ng, int B, int Gmax, float pos_iou_thre,
                              float neg_iou_thre, int* claim, float* offsets, int64_t* labels, float* matched_box, int64_t* matched_idx,
                              crog_stream_t stream) {
  CROG_CHECK_ARG(anchors && gt && ng && claim && offsets && labels && matched_box && matched_idx, "ssg_match: null pointer");
  CROG_CHECK_ARG(A > 0 && B > 0 && Gmax > 0 && B <= 65535, "ssg_match: bad sizes A=%d B=%d Gmax=%d", A, B, Gmax);
  hipLaunchKernelGGL(ssg_claim_kernel, dim3(Gmax, B), dim3(NT), 0, (hipStream_t)stream, anchors, A, gt, ng, Gmax, claim);
  hipLaunchKernelGGL(ssg_assign_kernel, dim3(cdiv(A, NT), B), dim3(NT), 0, (hipStream_t)stream, anchors, A, gt, ng, Gmax, claim, pos_iou_thre,
                     neg_iou_thre, offsets, labels, matched_box, matched_idx);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
