// Training-input preprocessing on the device: OCIDVLGDataset.preprocess (utils/dataset.py:843-914) for a batch
// of same-sized uint8 samples in ONE launch — letterbox warp of the image (cv2.warpAffine, INTER_CUBIC, CLIP-mean border) + CLIP
// normalisation, bilinear warp of the four uint8 target masks (instance, quality, angle, width), /255, degrees -> sin / cos (2 theta).
//
// The warp repeats OpenCV's 8-bit arithmetic (oracle/preprocess_oracle.py cites it): source coordinates in 10-bit fixed point cut to
// a 1/32-pixel grid, 15-bit fixed-point weight tables (built on the host, [1024][16] bicubic and [1024][4] bilinear, each entry
// summing to 2^15), constant border, (s + 2^14) >> 15 with saturation.  One thread per output pixel: 48 + 16 byte taps in, 8 floats
// out; HBM-bound (1.2 MB of uint8 in, 5.5 MB of fp32 out per 640 x 480 -> 416 x 416 sample), the tables stay in L1/L2.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NT = 256;

__device__ inline int tap_u8(const uint8_t* __restrict__ p, int H, int W, int C, int c, int y, int x, int cval) {
  return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? (int)p[((long)y * W + x) * C + c] : cval;
}
__device__ inline int fix_round(int acc) {      // FixedPtCast<int, uchar, 15>
  const int v = (acc + (1 << 14)) >> 15;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ void __launch_bounds__(NT) preprocess_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ masks, int B, int H, int W,
                                                        double m0, double m1, double m2, double m3, double m4, double m5,
                                                        const int16_t* __restrict__ tab_cubic, const int16_t* __restrict__ tab_linear, int S,
                                                        float mean0, float mean1, float mean2, float std0, float std1, float std2,
                                                        int bord0, int bord1, int bord2, float* __restrict__ out_img, float* __restrict__ out_masks) {
  const long n = (long)B * S * S;
  const long gid = (long)blockIdx.x * NT + threadIdx.x;
  if (gid >= n) return;
  const int x = (int)(gid % S), y = (int)((gid / S) % S), b = (int)(gid / ((long)S * S));
  // WarpAffineInvoker (imgwarp.cpp): AB_BITS = 10, INTER_BITS = 5, round_delta = 1024 / 32 / 2
  const int adelta = __double2int_rn(m0 * (double)x * 1024.0), bdelta = __double2int_rn(m3 * (double)x * 1024.0);
  const int X0 = __double2int_rn((m1 * (double)y + m2) * 1024.0) + 16, Y0 = __double2int_rn((m4 * (double)y + m5) * 1024.0) + 16;
  const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  const int sx = X >> 5, sy = Y >> 5;
  const int alpha = (Y & 31) * 32 + (X & 31);
  const long plane = (long)S * S, opix = (long)y * S + x;
  // image: bicubic, 4 x 4 taps from (sy - 1, sx - 1), three interleaved channels
  {
    const uint8_t* src = img + (long)b * H * W * 3;
    const int16_t* w = tab_cubic + alpha * 16;
    int acc[3] = {0, 0, 0};
    const int bord[3] = {bord0, bord1, bord2};
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int wt = w[r * 4 + c];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) acc[ch] += wt * tap_u8(src, H, W, 3, ch, sy - 1 + r, sx - 1 + c, bord[ch]);
      }
    const float mean[3] = {mean0, mean1, mean2}, sd[3] = {std0, std1, std2};
#pragma unroll
    for (int ch = 0; ch < 3; ch++)    // img.float().div_(255.).sub_(mean).div_(std)   (dataset.py:861-866)
      out_img[((long)b * 3 + ch) * plane + opix] = ((float)fix_round(acc[ch]) / 255.f - mean[ch]) / sd[ch];
  }
  // masks: bilinear, 2 x 2 taps from (sy, sx), border 0; planes: instance, quality, angle (degrees), width
  {
    const int16_t* w = tab_linear + alpha * 4;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint8_t* src = masks + ((long)b * 4 + k) * H * W;
      int acc = 0;
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) acc += (int)w[r * 2 + c] * tap_u8(src, H, W, 1, 0, sy + r, sx + c, 0);
      v[k] = fix_round(acc);
    }
    float* o = out_masks + (long)b * 5 * plane + opix;
    o[0] = (float)((double)v[0] / 255.0);                       // ins_mask / 255.            (dataset.py:890)
    o[plane] = (float)((double)v[1] / 255.0);                   // grasp_qua_mask / 255.
    const double theta = (double)v[2] * 3.141592653589793 / 180.0;      // grasp_ang_mask * np.pi / 180.
    o[2 * plane] = (float)sin(2.0 * theta);
    o[3 * plane] = (float)cos(2.0 * theta);
    o[4 * plane] = (float)((double)v[3] / 255.0);
  }
}

}  // namespace

extern "C" int crog_preprocess_u8(const uint8_t* img, const uint8_t* masks, int B, int H, int W, const double* minv, const int16_t* tab_cubic,
                                  const int16_t* tab_linear, int S, const float* mean, const float* stdv, const int* border, float* out_img,
                                  float* out_masks, crog_stream_t stream) {
  CROG_CHECK_ARG(img && masks && minv && tab_cubic && tab_linear && mean && stdv && border && out_img && out_masks, "preprocess: null pointer");
  CROG_CHECK_ARG(B > 0 && H > 0 && W > 0 && S > 0 && H < 32768 && W < 32768, "preprocess: bad sizes B=%d H=%d W=%d S=%d", B, H, W, S);
  const long n = (long)B * S * S;
  hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(n, NT)), dim3(NT), 0, (hipStream_t)stream, img, masks, B, H, W, minv[0], minv[1], minv[2], minv[3],
                     minv[4], minv[5], tab_cubic, tab_linear, S, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], border[0], border[1], border[2],
                     out_img, out_masks);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
