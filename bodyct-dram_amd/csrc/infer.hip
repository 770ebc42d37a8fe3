// Per-lobe whole-scan inference helpers, gfx950.
//
// Device-side restatement of the data movement around the model call in the reference's
// `LesionSegChunkTrain.evaluate_scan` (dram/job_runner.py:729-770) and of the thresholding in
// `LesionSegTest.run` (dram/job_runner.py:1003-1005, `binary_cam` dram/utils.py:226-242):
//   label_bboxes      bounding box of every lobe label            (find_crops, utils.py:244-254)
//   lobe_chunks       crop + mask outside lobe to -2048 + window to [0,1] + resample to R^3
//                     (job_runner.py:733-737, data_transforms.py:37-54, Resample 'fixed_size')
//   lobe_paste        sigmoid -> trilinear (align_corners) resize to the crop size -> paste where
//                     lobe == label (job_runner.py:765-770)
//   hist256 / threshold: 8-bit histogram inside the lungs for Otsu, mask = htp > th
// All HBM-bound gathers/streams; one thread per output voxel.  The scan (int16) and the lobe label
// map (uint8) stay resident in HBM; nothing goes back to the host except 5 boxes and 256 counts.
//
// Resampling grid: the reference resamples the crop with SimpleITK (absent here: parity unpinned); the crop -> R^3 step
// restates the grid of that ResampleImageFilter call from ITK's published semantics (itk_index below); the way back is the
// reference's own F.interpolate(..., align_corners=True) (job_runner.py:767).
#include "common.h"

namespace dram {

// boxes[l*6 + {0,1,2}] = min z,y,x ; {3,4,5} = max z,y,x (inclusive); min > max when the label is absent
__global__ void bbox_init_kernel(int* boxes, int nlabels) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nlabels * 6) boxes[i] = (i % 6) < 3 ? 0x7fffffff : -1;
}

// One block per (plane z, band of BBOX_ROWS rows): the band is one contiguous span of the label map, read 16 voxels per thread
// and step (rows of a multiple of 16 voxels on a 16-byte aligned map; one voxel per thread otherwise).  Every label 1..nlabels
// is tracked (evaluate_scan visits every value of np.unique(lobe)[1:], job_runner.py:729, not just 1..5): per-label x / y
// extents of the band in LDS -- an extent is only touched by an atomic when the voxel would widen it (a plain read first: the
// extents are monotonic, a stale read costs at most a redundant atomic), so that the uniform interior of a lobe costs a read
// per 16 voxels -- then one integer atomic set per label present in the band (deterministic).
// (Round 3 ran one block per ROW: 153,600 blocks of 512 bytes for a 300 x 512 x 512 map, 2.7 ms; this form: see DESIGN.md.)
constexpr int BBOX_ROWS = 64;
__device__ __forceinline__ void bbox_note(int* lx0, int* lx1, int* ly0, int* ly1, int lab, int xa, int xb, int y) {
    if (xa < lx0[lab]) atomicMin(&lx0[lab], xa);
    if (xb > lx1[lab]) atomicMax(&lx1[lab], xb);
    if (y < ly0[lab]) atomicMin(&ly0[lab], y);
    if (y > ly1[lab]) atomicMax(&ly1[lab], y);
}
__global__ __launch_bounds__(256) void label_bboxes_kernel(const uint8_t* __restrict__ lobe, int* __restrict__ boxes,
                                                           int nlabels, int D, int H, int W) {
    __shared__ int lx0[256], lx1[256], ly0[256], ly1[256];
    lx0[threadIdx.x] = 0x7fffffff; lx1[threadIdx.x] = -1;
    ly0[threadIdx.x] = 0x7fffffff; ly1[threadIdx.x] = -1;
    __syncthreads();
    const int z = blockIdx.y;
    const int y_lo = blockIdx.x * BBOX_ROWS;
    const int nrows = (H - y_lo) < BBOX_ROWS ? (H - y_lo) : BBOX_ROWS;
    const uint8_t* band = lobe + ((size_t)z * H + y_lo) * W;
    const int n = nrows * W;
    if ((W & 15) == 0 && (((size_t)band) & 15) == 0) {
        for (int e = 16 * (int)threadIdx.x; e < n; e += 16 * 256) {
            const uint4 v = *reinterpret_cast<const uint4*>(band + e);
            const int y = y_lo + e / W, x = e % W;
            const unsigned first = v.x & 0xffu;
            const bool uniform = v.x == first * 0x01010101u && v.y == v.x && v.z == v.x && v.w == v.x;
            if (uniform) {
                if (first != 0 && (int)first <= nlabels) bbox_note(lx0, lx1, ly0, ly1, (int)first, x, x + 15, y);
            } else {
                const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int lab = (int)((wds[k >> 2] >> (8 * (k & 3))) & 0xffu);
                    if (lab != 0 && lab <= nlabels) bbox_note(lx0, lx1, ly0, ly1, lab, x + k, x + k, y);
                }
            }
        }
    } else {
        for (int e = (int)threadIdx.x; e < n; e += 256) {
            const int lab = (int)band[e];
            if (lab != 0 && lab <= nlabels) bbox_note(lx0, lx1, ly0, ly1, lab, e % W, e % W, y_lo + e / W);
        }
    }
    __syncthreads();
    const int l = threadIdx.x;
    if (l >= 1 && l <= nlabels && lx1[l] >= 0) {
        int* bx = boxes + (l - 1) * 6;
        atomicMin(bx + 0, z); atomicMin(bx + 1, ly0[l]); atomicMin(bx + 2, lx0[l]);
        atomicMax(bx + 3, z); atomicMax(bx + 4, ly1[l]); atomicMax(bx + 5, lx1[l]);
    }
}

struct Chunk {   // crop window [z0,z0+dz) x [y0,y0+dy) x [x0,x0+dx) of lobe `label`
    int z0, y0, x0, dz, dy, dx, label, pad;
};
constexpr int MAX_CHUNKS = 8;
struct ChunkList {
    Chunk c[MAX_CHUNKS];
};

__device__ __forceinline__ void ac_index(int in, int out, int o, int& i0, int& i1, float& l0, float& l1) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;   // ATen area_pixel_compute_scale
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// The crop -> R^3 grid of the reference: Resample('fixed_size') (data_transforms.py:170-175) -> utils.resample ->
// sitk.ResampleImageFilter.Execute(image, new_size, identity transform, sitkLinear, the image's own origin and direction,
// new_spacing = spacing * size_in / size_out, fill 0) (utils.py:371-381).  SimpleITK is not available here (parity unpinned);
// the grid is restated from ITK's published semantics of that call: output voxel o lies at physical position origin +
// o * new_spacing, i.e. at continuous input index c = o * size_in / size_out; it is inside the input buffer while
// c < size_in - 0.5 (ImageFunction::IsInsideBuffer: [-0.5, size - 0.5)), otherwise the default value 0; linear interpolation
// between floor(c) and floor(c) + 1, the upper neighbour clamped to the last voxel (LinearInterpolateImageFunction).
__device__ __forceinline__ void itk_index(int in, int out, int o, int& i0, int& i1, float& l0, float& l1, bool& inside) {
    const double c = (double)o * (double)in / (double)out;
    inside = c < (double)in - 0.5;
    int b = (int)c;
    b = b > in - 1 ? in - 1 : b;
    i0 = b;
    i1 = b + 1 <= in - 1 ? b + 1 : b;
    l1 = i1 == i0 ? 0.f : (float)(c - (double)b);
    l0 = 1.f - l1;
}

// out[l][R][R][R]: windowed, lobe-masked crop resampled to R^3
__global__ __launch_bounds__(256) void lobe_chunks_kernel(const int16_t* __restrict__ scan,
                                                          const uint8_t* __restrict__ lobe, float* __restrict__ out,
                                                          ChunkList cl, int H, int W, int R, float wmin, float wmax) {
    const int l = blockIdx.y;
    const Chunk c = cl.c[l];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= R * R * R) return;
    const int xo = e % R, yo = (e / R) % R, zo = e / (R * R);
    int z0, z1, y0, y1, x0, x1;
    float a0, a1, b0, b1, c0, c1;
    bool inz, iny, inx;
    itk_index(c.dz, R, zo, z0, z1, a0, a1, inz);
    itk_index(c.dy, R, yo, y0, y1, b0, b1, iny);
    itk_index(c.dx, R, xo, x0, x1, c0, c1, inx);
    if (!(inz && iny && inx)) {                                   // beyond the crop's buffer (a crop smaller than R): default value
        out[(size_t)l * R * R * R + e] = 0.f;
        return;
    }
    const float inv = 1.f / (wmax - wmin);
    auto at = [&](int z, int y, int x) -> float {
        const size_t o = ((size_t)(c.z0 + z) * H + (c.y0 + y)) * W + (c.x0 + x);
        if (lobe[o] != c.label) return 0.f;                       // -2048 HU clips to the window minimum -> 0
        float v = (float)scan[o];
        v = fminf(fmaxf(v, wmin), wmax);
        return (v - wmin) * inv;                                  // windowing(), utils.py:189-198, to_span (0,1)
    };
    const float v = a0 * (b0 * (c0 * at(z0, y0, x0) + c1 * at(z0, y0, x1)) + b1 * (c0 * at(z0, y1, x0) + c1 * at(z0, y1, x1))) +
                    a1 * (b0 * (c0 * at(z1, y0, x0) + c1 * at(z1, y0, x1)) + b1 * (c0 * at(z1, y1, x0) + c1 * at(z1, y1, x1)));
    out[(size_t)l * R * R * R + e] = v;
}

// htp[v] = trilinear_ac(sigmoid(dense_l))(v) for every voxel v of the crop with lobe == label
__global__ __launch_bounds__(256) void lobe_paste_kernel(const float* __restrict__ dense,
                                                         const uint8_t* __restrict__ lobe, float* __restrict__ htp,
                                                         ChunkList cl, int H, int W, int R) {
    const int l = blockIdx.y;
    const Chunk c = cl.c[l];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= c.dz * c.dy * c.dx) return;
    const int x = e % c.dx, y = (e / c.dx) % c.dy, z = e / (c.dx * c.dy);
    const size_t o = ((size_t)(c.z0 + z) * H + (c.y0 + y)) * W + (c.x0 + x);
    if (lobe[o] != c.label) return;
    int z0, z1, y0, y1, x0, x1;
    float a0, a1, b0, b1, c0, c1;
    ac_index(R, c.dz, z, z0, z1, a0, a1);
    ac_index(R, c.dy, y, y0, y1, b0, b1);
    ac_index(R, c.dx, x, x0, x1, c0, c1);
    const float* p = dense + (size_t)l * R * R * R;
    auto at = [&](int zz, int yy, int xx) -> float {
        const float d = p[((size_t)zz * R + yy) * R + xx];
        return 1.f / (1.f + expf(-d));                            // F.sigmoid before the resize, job_runner.py:765
    };
    htp[o] = a0 * (b0 * (c0 * at(z0, y0, x0) + c1 * at(z0, y0, x1)) + b1 * (c0 * at(z0, y1, x0) + c1 * at(z0, y1, x1))) +
             a1 * (b0 * (c0 * at(z1, y0, x0) + c1 * at(z1, y0, x1)) + b1 * (c0 * at(z1, y1, x0) + c1 * at(z1, y1, x1)));
}

// hist[b] = #{v : lobe[v] > 0, uint8(clip(htp[v],0,1)*255) == b}; also sum of htp inside the lungs (fp64) and the count
__global__ __launch_bounds__(256) void lung_hist_kernel(const float* __restrict__ htp, const uint8_t* __restrict__ lobe,
                                                        unsigned long long* __restrict__ hist, double* __restrict__ sum,
                                                        size_t n) {
    __shared__ unsigned lh[256];
    __shared__ double lsum[4];
    lh[threadIdx.x] = 0;
    __syncthreads();
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (lobe[i] > 0) {
            const float v = htp[i];
            const float w = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
            atomicAdd(&lh[(int)w], 1u);
            s += (double)v;
        }
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) lsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
    if (threadIdx.x == 0) atomicAdd(sum, lsum[0] + lsum[1] + lsum[2] + lsum[3]);
}

__global__ void threshold_kernel(const float* __restrict__ htp, uint8_t* __restrict__ mask, float th, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) mask[i] = htp[i] > th ? 1 : 0;
}

// ---- the post-processing tail of LesionSegTest.run (dram/job_runner.py:1003-1012, 1033-1037)
// w_scan = windowing(scan, from_span=(wmin, wmax), to_span=(0, 1)) (utils.py:189-198: numpy clips the int16 scan, subtracts in
// integers and divides by float(wmax - wmin): fp64), and binary_cam's 8-bit view of it (utils.py:233: windowing(., (0, 1)) ->
// (w / 1.0) * 255 + 0 -> astype(uint8) truncates).  The same fp64 operations in the same order, so that bin and comparison
// are bit-identical to numpy's.
__device__ __forceinline__ double windowed_scan(int s, int wmin, int wmax) {
    const int c = s < wmin ? wmin : (s > wmax ? wmax : s);
    return (double)(c - wmin) / (double)(wmax - wmin);
}

// hist[b] = #{v : lobe[v] > 0, uint8(w_scan[v] * 255) == b}
__global__ __launch_bounds__(256) void scan_hist_kernel(const int16_t* __restrict__ scan, const uint8_t* __restrict__ lobe,
                                                        unsigned long long* __restrict__ hist, int wmin, int wmax, size_t n) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (lobe[i] > 0) {
            const double w = windowed_scan((int)scan[i], wmin, wmax);
            atomicAdd(&lh[(int)((w / 1.0) * 255.0 + 0.0)], 1u);
        }
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// lesion_pred = htp > th (job_runner.py:1004);  lesion_pred_post = lesion_pred & (w_scan > th_scan) & ~(vessel > 0) (1008-1010)
__global__ void lesion_post_kernel(const float* __restrict__ htp, const int16_t* __restrict__ scan,
                                   const uint8_t* __restrict__ vessel, uint8_t* __restrict__ pred, uint8_t* __restrict__ post,
                                   float th, int wmin, int wmax, double th_scan, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool p = htp[i] > th;
        const bool bright = windowed_scan((int)scan[i], wmin, wmax) > th_scan;
        const bool ves = vessel != nullptr && vessel[i] > 0;
        if (pred) pred[i] = p ? 1 : 0;
        post[i] = (p && bright && !ves) ? 1 : 0;
    }
}

// counts = {|a & b|, |a | b|, |a|, |b|} of two masks (non-zero = set): what IOU / Dice (utils.py:437-446) are made of
__global__ __launch_bounds__(256) void mask_overlap_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                           unsigned long long* __restrict__ counts, size_t n) {
    unsigned c[4] = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const bool x = a[i] > 0, y = b[i] > 0;
        c[0] += x && y; c[1] += x || y; c[2] += x; c[3] += y;
    }
    __shared__ unsigned sm[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned v = c[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&counts[threadIdx.x], (unsigned long long)sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}


// ---- utils.resample(narray, spacing, required_spacing=..., new_size=..., interpolator=...) (utils.py:414-434 -> 299-381): the
// volumes of LesionSegTest.run going back to the scan's original grid (job_runner.py:1016-1032).  sitk.ResampleImageFilter with
// the identity transform, the image's own origin and direction, output spacing = required_spacing, default value 0, output
// pixel type = input pixel type.  SimpleITK 1.1.0 is not available (PARITY UNPINNED); restated from ITK's published semantics:
// output voxel o sits at continuous input index c = o * spacing_out / spacing_in per axis; it is inside the buffer while
// c < size_in - 0.5 (ImageFunction::IsInsideBuffer; c >= 0 always: same origin), else the default value.
//   nearest: voxel floor(c + 0.5) (NearestNeighborInterpolateImageFunction: Math::RoundHalfIntegerUp);
//   linear:  base = floor(c), upper neighbour clamped to the last voxel, lerps x, then y, then z in double
//            (LinearInterpolateImageFunction::EvaluateOptimized: RealType of short / float pixels is double), then the cast of
//            ResampleImageFilter::CastPixelWithBoundsChecking: clamp to the pixel type's range, static_cast (int16: truncation).
struct ResampleGrid {
    int Di, Hi, Wi, Do, Ho, Wo;
    double sz, sy, sx;          // spacing_out / spacing_in per axis
};
__device__ __forceinline__ double mul_rn(double a, double b) {       // (HIP's __dmul_rn is a plain `*`, open to contraction)
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ bool itk_axis(double c, int in, int& i0, int& i1, double& t) {
#pragma clang fp contract(off)
    const int b = (int)c;                                  // c >= 0
    i0 = b > in - 1 ? in - 1 : b;
    i1 = i0 + 1 <= in - 1 ? i0 + 1 : i0;
    t = i1 == i0 ? 0.0 : c - (double)i0;
    return c < (double)in - 0.5;
}
template <typename T> __device__ __forceinline__ T itk_cast(double v);
template <> __device__ __forceinline__ float itk_cast<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ int16_t itk_cast<int16_t>(double v) {
    return v < -32768.0 ? (int16_t)-32768 : v > 32767.0 ? (int16_t)32767 : (int16_t)v;
}
template <> __device__ __forceinline__ uint8_t itk_cast<uint8_t>(double v) {
    return v < 0.0 ? (uint8_t)0 : v > 255.0 ? (uint8_t)255 : (uint8_t)v;
}
template <typename T, bool LINEAR>
__global__ __launch_bounds__(256) void resample_volume_kernel(const T* __restrict__ in, T* __restrict__ out, ResampleGrid g) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= g.Wo) return;
    const size_t o = ((size_t)z * g.Ho + y) * g.Wo + x;
    // (products rounded on their own: hipcc would otherwise contract c - base into fma(o, s, -base), which is not what ITK or the oracle compute)
    const double cz = mul_rn((double)z, g.sz), cy = mul_rn((double)y, g.sy), cx = mul_rn((double)x, g.sx);
    int z0, z1, y0, y1, x0, x1;
    double tz, ty, tx;
    const bool inz = itk_axis(cz, g.Di, z0, z1, tz), iny = itk_axis(cy, g.Hi, y0, y1, ty), inx = itk_axis(cx, g.Wi, x0, x1, tx);
    if (!(inz && iny && inx)) { out[o] = (T)0; return; }
    auto at = [&](int zz, int yy, int xx) { return in[((size_t)zz * g.Hi + yy) * g.Wi + xx]; };
    if (!LINEAR) {
        out[o] = at((int)(cz + 0.5), (int)(cy + 0.5), (int)(cx + 0.5));       // (inside: c + 0.5 < size_in)
        return;
    }
    // (no fused multiply-add: ITK's x86 builds round the product, and so does the oracle -- it shows where an int16 result
    //  truncates: -484 + 705 * 0.4 is -202 with a rounded product, -201.99999999999997 fused.  HIP's __dmul_rn is a plain `*`.)
    auto lerp = [](double a, double b, double t) {
#pragma clang fp contract(off)
        const double p = (b - a) * t;
        return a + p;
    };
    const double v00 = lerp((double)at(z0, y0, x0), (double)at(z0, y0, x1), tx), v10 = lerp((double)at(z0, y1, x0), (double)at(z0, y1, x1), tx);
    const double v01 = lerp((double)at(z1, y0, x0), (double)at(z1, y0, x1), tx), v11 = lerp((double)at(z1, y1, x0), (double)at(z1, y1, x1), tx);
    out[o] = itk_cast<T>(lerp(lerp(v00, v10, ty), lerp(v01, v11, ty), tz));
}

}  // namespace dram

using namespace dram;

extern "C" int dram_label_bboxes(const uint8_t* lobe, int* boxes, int nlabels, int D, int H, int W, void* stream) {
    DRAM_REQUIRE(lobe && boxes, "label_bboxes: null pointer");
    DRAM_REQUIRE(nlabels > 0 && nlabels <= 255 && D > 0 && H > 0 && W > 0 && D <= 65535, "label_bboxes: bad dimensions");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(cdiv(nlabels * 6, 64)), dim3(64), 0, st, boxes, nlabels);
    hipLaunchKernelGGL(label_bboxes_kernel, dim3(cdiv(H, BBOX_ROWS), D), dim3(256), 0, st, lobe, boxes, nlabels, D, H, W);
    return check_launch("label_bboxes");
}

static int fill_chunks(const char* who, ChunkList& cl, const int* chunks, int L, int D, int H, int W) {
    DRAM_REQUIRE(chunks && L > 0 && L <= MAX_CHUNKS, "%s: between 1 and %d chunks", who, MAX_CHUNKS);
    for (int l = 0; l < L; ++l) {
        Chunk& c = cl.c[l];
        c.z0 = chunks[l * 7 + 0]; c.y0 = chunks[l * 7 + 1]; c.x0 = chunks[l * 7 + 2];
        c.dz = chunks[l * 7 + 3]; c.dy = chunks[l * 7 + 4]; c.dx = chunks[l * 7 + 5];
        c.label = chunks[l * 7 + 6]; c.pad = 0;
        DRAM_REQUIRE(c.z0 >= 0 && c.y0 >= 0 && c.x0 >= 0 && c.dz > 0 && c.dy > 0 && c.dx > 0 && c.z0 + c.dz <= D &&
                         c.y0 + c.dy <= H && c.x0 + c.dx <= W, "%s: chunk %d outside the volume", who, l);
    }
    return DRAM_OK;
}

// chunks: host array of L x {z0,y0,x0,dz,dy,dx,label}
extern "C" int dram_lobe_chunks(const int16_t* scan, const uint8_t* lobe, float* out, const int* chunks, int L, int D,
                                int H, int W, int R, float wmin, float wmax, void* stream) {
    DRAM_REQUIRE(scan && lobe && out, "lobe_chunks: null pointer");
    DRAM_REQUIRE(R > 0 && wmax > wmin, "lobe_chunks: bad resample size or window");
    ChunkList cl;
    int rc = fill_chunks("lobe_chunks", cl, chunks, L, D, H, W);
    if (rc) return rc;
    hipLaunchKernelGGL(lobe_chunks_kernel, dim3(cdiv(R * R * R, 256), L), dim3(256), 0, (hipStream_t)stream, scan, lobe,
                       out, cl, H, W, R, wmin, wmax);
    return check_launch("lobe_chunks");
}

extern "C" int dram_lobe_paste(const float* dense, const uint8_t* lobe, float* htp, const int* chunks, int L, int D,
                               int H, int W, int R, void* stream) {
    DRAM_REQUIRE(dense && lobe && htp, "lobe_paste: null pointer");
    DRAM_REQUIRE(R > 0, "lobe_paste: bad resample size");
    ChunkList cl;
    int rc = fill_chunks("lobe_paste", cl, chunks, L, D, H, W);
    if (rc) return rc;
    int64_t mx = 0;
    for (int l = 0; l < L; ++l) {
        const int64_t v = (int64_t)cl.c[l].dz * cl.c[l].dy * cl.c[l].dx;
        mx = v > mx ? v : mx;
    }
    DRAM_REQUIRE(mx < 0x7fffffffLL, "lobe_paste: chunk too large");
    hipLaunchKernelGGL(lobe_paste_kernel, dim3((unsigned)cdiv64(mx, 256), L), dim3(256), 0, (hipStream_t)stream, dense,
                       lobe, htp, cl, H, W, R);
    return check_launch("lobe_paste");
}

// hist: 256 x uint64 (zeroed here), sum: 1 x double (zeroed here)
extern "C" int dram_lung_hist256(const float* htp, const uint8_t* lobe, unsigned long long* hist, double* sum,
                                 int64_t n, void* stream) {
    DRAM_REQUIRE(htp && lobe && hist && sum && n > 0, "lung_hist256: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(hist, 0, 256 * sizeof(unsigned long long), st);
    (void)hipMemsetAsync(sum, 0, sizeof(double), st);
    const unsigned grid = (unsigned)(cdiv64(n, 256 * 16) < 2048 ? cdiv64(n, 256 * 16) : 2048);
    hipLaunchKernelGGL(lung_hist_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, htp, lobe, hist, sum, (size_t)n);
    return check_launch("lung_hist256");
}

extern "C" int dram_threshold_mask(const float* htp, uint8_t* mask, float th, int64_t n, void* stream) {
    DRAM_REQUIRE(htp && mask && n > 0, "threshold_mask: bad arguments");
    const unsigned grid = (unsigned)(cdiv64(n, 256) < 4096 ? cdiv64(n, 256) : 4096);
    hipLaunchKernelGGL(threshold_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, htp, mask, th, (size_t)n);
    return check_launch("threshold_mask");
}

// hist: 256 x uint64 (zeroed here) of binary_cam's 8-bit view of windowing(scan, (wmin, wmax), (0, 1)) inside the lungs
extern "C" int dram_scan_hist256(const int16_t* scan, const uint8_t* lobe, unsigned long long* hist, int wmin, int wmax,
                                 int64_t n, void* stream) {
    DRAM_REQUIRE(scan && lobe && hist && n > 0 && wmax > wmin, "scan_hist256: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(hist, 0, 256 * sizeof(unsigned long long), st);
    const unsigned grid = (unsigned)(cdiv64(n, 256 * 16) < 2048 ? cdiv64(n, 256 * 16) : 2048);
    hipLaunchKernelGGL(scan_hist_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, scan, lobe, hist, wmin, wmax, (size_t)n);
    return check_launch("scan_hist256");
}

// pred (may be NULL) = htp > th;  post = pred & (windowing(scan) > th_scan) & ~(vessel > 0)  (vessel may be NULL: no vessels)
extern "C" int dram_lesion_post(const float* htp, const int16_t* scan, const uint8_t* vessel, uint8_t* pred, uint8_t* post,
                                float th, int wmin, int wmax, double th_scan, int64_t n, void* stream) {
    DRAM_REQUIRE(htp && scan && post && n > 0 && wmax > wmin, "lesion_post: bad arguments");
    const unsigned grid = (unsigned)(cdiv64(n, 256) < 4096 ? cdiv64(n, 256) : 4096);
    hipLaunchKernelGGL(lesion_post_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, htp, scan, vessel, pred, post, th,
                       wmin, wmax, th_scan, (size_t)n);
    return check_launch("lesion_post");
}

// counts[4] (uint64, zeroed here) = {intersection, union, |a|, |b|}
extern "C" int dram_mask_overlap(const uint8_t* a, const uint8_t* b, unsigned long long* counts, int64_t n, void* stream) {
    DRAM_REQUIRE(a && b && counts && n > 0, "mask_overlap: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), st);
    const unsigned grid = (unsigned)(cdiv64(n, 256 * 16) < 2048 ? cdiv64(n, 256 * 16) : 2048);
    hipLaunchKernelGGL(mask_overlap_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, a, b, counts, (size_t)n);
    return check_launch("mask_overlap");
}

// kind: 0 = uint8, 1 = int16, 2 = float32 volumes; linear: 0 = nearest neighbour, 1 = linear.  spacing_*: (z, y, x).
extern "C" int dram_resample_volume(const void* in, void* out, int kind, int linear, int Di, int Hi, int Wi, int Do, int Ho,
                                    int Wo, const double* spacing_in, const double* spacing_out, void* stream) {
    DRAM_REQUIRE(in && out && spacing_in && spacing_out, "resample_volume: null pointer");
    DRAM_REQUIRE(kind >= 0 && kind <= 2 && (linear == 0 || linear == 1), "resample_volume: kind in 0..2, linear in 0..1");
    DRAM_REQUIRE(Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && Ho <= 65535 && Do <= 65535,
                 "resample_volume: bad dimensions");
    ResampleGrid g{Di, Hi, Wi, Do, Ho, Wo, 0.0, 0.0, 0.0};
    for (int a = 0; a < 3; ++a) DRAM_REQUIRE(spacing_in[a] > 0.0 && spacing_out[a] > 0.0, "resample_volume: spacings must be positive");
    g.sz = spacing_out[0] / spacing_in[0]; g.sy = spacing_out[1] / spacing_in[1]; g.sx = spacing_out[2] / spacing_in[2];
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv(Wo, 256), Ho, Do), block(256);
#define DRAM_RS(T_, L_) hipLaunchKernelGGL((resample_volume_kernel<T_, L_>), grid, block, 0, st, (const T_*)in, (T_*)out, g)
    if (kind == 0) { if (linear) DRAM_RS(uint8_t, true); else DRAM_RS(uint8_t, false); }
    else if (kind == 1) { if (linear) DRAM_RS(int16_t, true); else DRAM_RS(int16_t, false); }
    else { if (linear) DRAM_RS(float, true); else DRAM_RS(float, false); }
#undef DRAM_RS
    return check_launch("resample_volume");
}
