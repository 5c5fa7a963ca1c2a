// wh_mel.hip — fused STFT + power + mel filterbank + log10 (+ global max) for gfx950.
//
// Replaces whisper_log_mel_80 (reference src/main.rs:407-509) and its rustfft 400-point FFT
// (src/main.rs:440-441,473).  One workgroup = 16 consecutive frames of one clip:
//   1. the 2800 samples those frames touch are staged once in LDS, reflect padding applied by
//      index arithmetic (src/main.rs:419-435) — no padded copy of the clip exists;
//   2. the 400-point real DFT is a mixed-radix FFT in LDS: a 200-point complex Stockham FFT (radix 5, 5, 8) of
//      z[n] = xw[2n] + i xw[2n+1] followed by the real-input untangle (xw = sample * periodic-Hann in f32 exactly as
//      src/main.rs:463-470), f32 arithmetic as in the reference's rustfft.  Twiddles come from one 400-entry cosine
//      table in LDS (computed in f64, rounded once);
//   3. power = fl32(re)^2 + fl32(im)^2 (src/main.rs:476-481) goes to LDS and the 201 -> n_mels
//      filterbank contraction runs on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32, a
//      k-ordered f32 fma chain like the reference's scalar loop, src/main.rs:484-490);
//   4. max(.,1e-10), log10, coalesced store of raw log-power [n_mels][frames], and a per-clip
//      global max via one atomicMax per workgroup (src/main.rs:494-500).
// k_mel_norm / k_mel_tokens then apply (max(x, gmax-8)+4)/4 (src/main.rs:502-506) while writing
// either the caller-visible [n_mels][frames] f32 array or the token-major conv1 operand.
//
// Algorithmic HBM bytes per 30 s clip: 1.92 MB PCM in + 0.96 MB raw log-mel out.
#include "wh_common.h"
#include "wh_kernels.h"

#include <math.h>
#include <vector>

namespace {

constexpr int N_FFT = 400;
constexpr int HOP = 160;
constexpr int N_FREQ = 201;
constexpr int NBIN_PAD = 208;  // 13 tiles of 16
// frames per workgroup (FRB, template parameter): 16 = one full MFMA tile of the filterbank (2800 staged samples, 65 KB of
// LDS, two workgroups per CU).  8 frames (1520 samples, 34 KB, four workgroups per CU: four independent barrier-separated
// latency chains per CU instead of two) was measured in round 3 and is slower — 7.28 against 5.21 ms per 1024 clips: the same
// waves per CU, but half of every filterbank MFMA tile idle and twice the per-workgroup fixed cost (tables, barriers).

// padded-signal sample i of a clip with n samples (src/main.rs:419-435)
__device__ __forceinline__ float padded_sample(const float* __restrict__ pcm, long n, long i) {
    if (n < 2) return (i < n) ? pcm[i] : 0.0f;  // extend_from_slice, then zero-resize
    if (i < 200) {
        long idx = 200 - i;
        return pcm[idx < n - 1 ? idx : n - 1];
    }
    long j = i - 200;
    if (j < n) return pcm[j];
    j -= n;  // 0..199
    if (j >= 200) return 0.0f;
    long idx = n - 2 - j;
    return pcm[idx > 0 ? idx : 0];
}

constexpr int N_HALF = N_FFT / 2;   // length of the complex FFT behind the real transform
typedef float real_t;   // FFT arithmetic type (see the accuracy note at the FFT)
typedef __attribute__((ext_vector_type(2))) real_t cplx;   // (re, im)

// a * e^(-i theta), theta = 2 pi t / 400: cos from the table, sin x = cos(x - pi/2)
__device__ __forceinline__ cplx cmul_tw(const cplx a, const real_t* tw, int t) {
    const real_t c = tw[t], s = tw[t >= 100 ? t - 100 : t + 300];
    return cplx{a.x * c + a.y * s, a.y * c - a.x * s};
}
__device__ __forceinline__ cplx mul_mi(const cplx z) { return cplx{z.y, -z.x}; }   // z * (-i)

// forward 5-point DFT (e^(-2 pi i u v/5))
__device__ __forceinline__ void dft5(const cplx (&a)[5], cplx (&y)[5]) {
    constexpr real_t c1 = (real_t)0.30901699437494742410, c2 = (real_t)-0.80901699437494742410;   // cos 2pi/5, cos 4pi/5
    constexpr real_t s1 = (real_t)0.95105651629515357212, s2 = (real_t)0.58778525229247312917;    // sin 2pi/5, sin 4pi/5
    const cplx t1 = a[1] + a[4], t2 = a[2] + a[3], t3 = a[1] - a[4], t4 = a[2] - a[3];
    y[0] = a[0] + t1 + t2;
    const cplx m1 = a[0] + c1 * t1 + c2 * t2, m2 = a[0] + c2 * t1 + c1 * t2;
    const cplx n1 = mul_mi(s1 * t3 + s2 * t4), n2 = mul_mi(s2 * t3 - s1 * t4);
    y[1] = m1 + n1; y[4] = m1 - n1;
    y[2] = m2 + n2; y[3] = m2 - n2;
}
// forward 8-point DFT
__device__ __forceinline__ void dft8(const cplx (&a)[8], cplx (&y)[8]) {
    constexpr real_t r = (real_t)0.70710678118654752440;
    const cplx b0 = a[0] + a[4], b1 = a[0] - a[4], b2 = a[2] + a[6], b3 = a[2] - a[6];
    const cplx b4 = a[1] + a[5], b5 = a[1] - a[5], b6 = a[3] + a[7], b7 = a[3] - a[7];
    const cplx c0 = b0 + b2, c1 = b0 - b2, c2 = b1 + mul_mi(b3), c3 = b1 - mul_mi(b3);
    const cplx c4 = b4 + b6, c5 = b4 - b6, c6 = b5 + mul_mi(b7), c7 = b5 - mul_mi(b7);
    const cplx w6 = cplx{(c6.x + c6.y) * r, (c6.y - c6.x) * r};      // c6 * (1 - i)/sqrt 2
    const cplx w7 = cplx{(c7.y - c7.x) * r, (-c7.x - c7.y) * r};     // c7 * (-1 - i)/sqrt 2
    y[0] = c0 + c4; y[4] = c0 - c4;
    y[2] = c1 + mul_mi(c5); y[6] = c1 - mul_mi(c5);
    y[1] = c2 + w6; y[5] = c2 - w6;
    y[3] = c3 + w7; y[7] = c3 - w7;
}

// FRB frames x 32 threads: the FFT stages are latency chains, more lanes per frame shorten them
template <int FRB>
__global__ __launch_bounds__(FRB * 32) void k_mel_stft(const float* __restrict__ pcm, long pcm_stride,
                                                  const int* __restrict__ n_samples, const double* __restrict__ tw_g,
                                                  const float* __restrict__ win_g, const float* __restrict__ fbT,
                                                  int n_mels, float* __restrict__ raw, long raw_clip_stride,
                                                  long raw_row_stride, unsigned* __restrict__ gmax) {
    constexpr int FR_BLK = FRB, MEL_THREADS = FRB * 32, SPAN = (FRB - 1) * HOP + N_FFT;
    __shared__ __attribute__((aligned(16))) real_t tw[N_FFT];
    __shared__ __attribute__((aligned(16))) float smp[SPAN];
    __shared__ __attribute__((aligned(16))) float win[N_FFT];

    const int clip = blockIdx.y;
    const long n = n_samples[clip];
    const long n_frames = (n / HOP) > 1 ? (n / HOP) : 1;  // src/main.rs:444-452
    const long f0 = (long)blockIdx.x * FR_BLK;
    if (f0 >= n_frames) return;
    const float* cp = pcm + (long)clip * pcm_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < N_FFT; i += MEL_THREADS) {
        tw[i] = (real_t)tw_g[i];
        win[i] = win_g[i];
    }
    {   // all of a thread's samples in flight before the LDS stores
        constexpr int NS = (SPAN + MEL_THREADS - 1) / MEL_THREADS;
        float sv[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) sv[i] = padded_sample(cp, n, f0 * HOP + min(tid + i * MEL_THREADS, SPAN - 1));
#pragma unroll
        for (int i = 0; i < NS; i++)
            if (tid + i * MEL_THREADS < SPAN) smp[tid + i * MEL_THREADS] = sv[i];
    }
    __syncthreads();

    // ---- 400-point real DFT of 16 frames: f64 mixed-radix FFT in LDS --------------------------------------------
    // z[n] = xw[2n] + i xw[2n+1] (xw = sample * periodic Hann in f32 exactly as src/main.rs:463-470), a 200-point complex
    // Stockham FFT (radix 5, 5, 8: thread task i of a stage reads in[i + u*T], multiplies by w^(u*k), k = i mod p, and
    // writes the R-point DFT to out[(i-k)*R + k + v*p]), then the real-input untangle
    //     X[k] = (Z[k] + conj Z[200-k]) / 2  -  i e^(-2 pi i k/400) (Z[k] - conj Z[200-k]) / 2,   k = 0..200.
    // Every twiddle is a multiple of 2 pi/400 and comes from the one 400-entry f64 cosine table (sin x = cos(x - pi/2)).
    // ~12 kflop per frame on the vector ALUs instead of the 160 kflop of a dense DFT on the f64 matrix cores (which bound
    // the previous version of this kernel at 3.3 ms per 256 clips).  Arithmetic in f32 like the reference's rustfft
    // (src/main.rs:440-441,473 run an f32 FFT): the spectrum's error is ~1e-7 of the frame's largest bin, so a bin at the
    // floor of the 8-decade log range (1e-4 of the peak amplitude) carries <= 1e-3 relative error = 1e-4 in the normalised
    // log-mel; measured against the f64-DFT oracle every test clip stays within the 1e-4 tolerance, as the f64 variant of
    // this FFT did (2.1 ms; f32 halves the LDS footprint: two workgroups per CU).
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    cplx* bufA = reinterpret_cast<cplx*>(dyn_smem);            // [FR_BLK][200]
    cplx* bufB = bufA + FR_BLK * N_HALF;                       // [FR_BLK][200]
    float (*pw)[NBIN_PAD] = reinterpret_cast<float (*)[NBIN_PAD]>(bufB);   // power spectrum: reuses bufB, dead after stage 3
    // task mapping without integer division: thread = (frame tid >> 5, slot tid & 31), a frame's tasks are walked 32 at a time
    const int ff = tid >> 5, sl = tid & 31;
    {   // stage 1: radix 5, p = 1, T = 40 (no twiddles); inputs straight from the staged samples
        const float* sf = smp + ff * HOP;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int i = sl + 32 * it;
            if (i < 40) {
                cplx a[5];
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    const int n2 = 2 * (i + 40 * u);
                    a[u] = cplx{(real_t)(sf[n2] * win[n2]), (real_t)(sf[n2 + 1] * win[n2 + 1])};
                }
                cplx y[5];
                dft5(a, y);
                cplx* o = bufA + ff * N_HALF + i * 5;
#pragma unroll
                for (int v = 0; v < 5; v++) o[v] = y[v];
            }
        }
    }
    __syncthreads();
    {   // stage 2: radix 5, p = 5: twiddle e^(-2 pi i u k/25) = table index 16 u k (<= 256: no wrap)
        const cplx* in = bufA + ff * N_HALF;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int i = sl + 32 * it;
            if (i < 40) {
                const int k = i - 5 * ((i * 13) >> 6);   // i mod 5 for i < 64
                cplx a[5];
#pragma unroll
                for (int u = 0; u < 5; u++) a[u] = cmul_tw(in[i + 40 * u], tw, 16 * u * k);
                cplx y[5];
                dft5(a, y);
                cplx* o = bufB + ff * N_HALF + (i - k) * 5 + k;
#pragma unroll
                for (int v = 0; v < 5; v++) o[5 * v] = y[v];
            }
        }
    }
    __syncthreads();
    {   // stage 3: radix 8, p = 25, T = 25: twiddle e^(-2 pi i u k/200) = table index 2 u k (<= 336: no wrap)
        const cplx* in = bufB + ff * N_HALF;
        {
            const int k = sl;
            if (k < 25) {
                cplx a[8];
#pragma unroll
                for (int u = 0; u < 8; u++) a[u] = cmul_tw(in[k + 25 * u], tw, 2 * u * k);
                cplx y[8];
                dft8(a, y);
                cplx* o = bufA + ff * N_HALF + k;
#pragma unroll
                for (int v = 0; v < 8; v++) o[25 * v] = y[v];
            }
        }
    }
    __syncthreads();
    // ---- untangle + power spectrum to LDS (src/main.rs:476-481: re, im as f32, then re^2 + im^2) ------------------
    {
        const cplx* Z = bufA + ff * N_HALF;
#pragma unroll
        for (int it = 0; it < (NBIN_PAD + 31) / 32; it++) {
            const int k = sl + 32 * it;
            if (k >= NBIN_PAD) break;
            float p = 0.0f;
            if (k <= 200) {
                const cplx A = Z[k == 200 ? 0 : k], Bc = Z[k == 0 ? 0 : 200 - k];
                const real_t ex = (real_t)0.5 * (A.x + Bc.x), ey = (real_t)0.5 * (A.y - Bc.y);     // E = (A + conj B) / 2
                const real_t dx = (real_t)0.5 * (A.x - Bc.x), dy = (real_t)0.5 * (A.y + Bc.y);     // (A - conj B) / 2
                const real_t ox = dy, oy = -dx;                                                        // O = -i (A - conj B) / 2
                const real_t c = tw[k], sn = tw[k >= 100 ? k - 100 : k + 300];                         // e^(-2 pi i k/400) = c - i sn
                const float re = (float)(ex + c * ox + sn * oy), im = (float)(ey + c * oy - sn * ox);
                p = re * re + im * im;
            }
            pw[ff][k] = p;
        }
    }
    __syncthreads();

    // ---- mel filterbank on the exact-f32 matrix cores ------------------------------------------
    const int fl = lane & 15, g = lane >> 4;
    // A[i = frame][k = bin] = pw, B[k = bin][j = mel] = fbT[bin][mel]; D: col j = mel (lane&15),
    // rows i = frame 4*(lane>>4)+r  → each lane owns 4 consecutive frames of one mel row.
    float lmax = -INFINITY;
    const int n_mt = n_mels >> 4;
    for (int mt = wave; mt < n_mt; mt += MEL_THREADS / 64) {
        f32x4 acc = {0, 0, 0, 0};
        const int mel = 16 * mt + fl;
#pragma unroll 13
        for (int step = 0; step < NBIN_PAD / 4; step++) {
            const int k = 4 * step + g;
            float a = pw[fl & (FR_BLK - 1)][k];   // (FRB = 8: rows 8..15 of the tile repeat rows 0..7 and are not stored)
            float b = fbT[(long)k * n_mels + mel];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        float* out = raw + (long)clip * raw_clip_stride + (long)mel * raw_row_stride + f0 + 4 * g;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = log10f(fmaxf(acc[r], 1e-10f));  // src/main.rs:489,496
            if (4 * g < FR_BLK && f0 + 4 * g + r < n_frames) lmax = fmaxf(lmax, v[r]);
        }
        if (4 * g >= FR_BLK) {
            // (FRB = 8: the upper half of the tile belongs to the next workgroup)
        } else if (f0 + 4 * g + 3 < n_frames) {
            store4(out, v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (f0 + 4 * g + r < n_frames) out[r] = v[r];
        }
    }
    lmax = dpp_wave_max(lmax);
    if (lane == 0 && lmax > -INFINITY) atomicMax(gmax + clip, f2ord(lmax));
}

// (max(x, gmax-8)+4)/4 into the caller-visible [n_mels][n_frames] array
__global__ void k_mel_norm(const float* __restrict__ raw, long raw_row_stride, const unsigned* __restrict__ gmax,
                           int n_mels, long n_frames, float* __restrict__ out) {
    const float mx = ord2f(gmax[0]);
    const long tot = (long)n_mels * n_frames;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        long m = i / n_frames, f = i - m * n_frames;
        float lv = raw[m * raw_row_stride + f];
        out[i] = (fmaxf(lv, mx - 8.0f) + 4.0f) * 0.25f;
    }
}

// Window of 3000 frames starting at frame_start of source clip src[b], normalised, zero-filled past
// the clip's last frame (src/main.rs:895-905), transposed to token-major rows for conv1:
//   tok[b][1 + t][m]  (row 0 and row 3001 are the conv's zero padding and are left untouched).
// mode 0: src is raw log-power + gmax; mode 1: src is already-normalised mel (wh_encode input).
template <typename T>
__global__ __launch_bounds__(256) void k_mel_tokens(const float* __restrict__ src, long src_clip_stride,
                                                    long src_row_stride, const int* __restrict__ src_index,
                                                    const int* __restrict__ frame_start,
                                                    const int* __restrict__ n_frames_src,
                                                    const unsigned* __restrict__ gmax, int mode, int n_mels,
                                                    T* __restrict__ tok, long tok_clip_stride) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y;
    const int sc = src_index ? src_index[b] : b;
    const long fs = frame_start ? frame_start[b] : 0;
    const long nf = n_frames_src[sc];
    const float mx = (mode == 0) ? ord2f(gmax[sc]) : 0.0f;
    const int t0 = blockIdx.x * 64;
    const float* sp = src + (long)sc * src_clip_stride;
    T* tp = tok + (long)b * tok_clip_stride;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int m0 = 0; m0 < n_mels; m0 += 64) {
        for (int r = ty; r < 64; r += 4) {
            int m = m0 + r;
            long f = fs + t0 + tx;
            float v = 0.0f;
            if (m < n_mels && t0 + tx < WH_N_FRAMES && f < nf) {
                float lv = sp[(long)m * src_row_stride + f];
                v = (mode == 0) ? (fmaxf(lv, mx - 8.0f) + 4.0f) * 0.25f : lv;
            }
            tile[r][tx] = v;
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            int t = t0 + r, m = m0 + tx;
            if (t < WH_N_FRAMES && m < n_mels) tp[(long)(1 + t) * n_mels + m] = cvt_out<T>(tile[tx][r]);
        }
        __syncthreads();
    }
}

// ---- host-side tables ---------------------------------------------------------------------------
float hz_to_mel(float hz) {  // src/main.rs:332-341
    const float logstep = 27.0f / logf(6.4f);
    float mel = 3.0f * hz / 200.0f;
    if (hz >= 1000.0f) mel = 15.0f + logf(hz / 1000.0f) * logstep;
    return mel;
}
float mel_to_hz(float mel) {  // src/main.rs:343-352
    const float logstep = logf(6.4f) / 27.0f;
    float hz = 200.0f * mel / 3.0f;
    if (mel >= 15.0f) hz = 1000.0f * expf(logstep * (mel - 15.0f));
    return hz;
}

}  // namespace

// Slaney filterbank in f32 (src/main.rs:354-405), stored transposed + zero padded: fbT[208][n_mels]
void wh_build_mel_tables(int n_mels, std::vector<double>& tw, std::vector<float>& win, std::vector<float>& fbT) {
    tw.resize(N_FFT);
    for (int j = 0; j < N_FFT; j++) tw[j] = cos(2.0 * M_PI * (double)j / N_FFT);
    win.resize(N_FFT);
    for (int i = 0; i < N_FFT; i++) {  // src/main.rs:323-330
        float x = (3.14159265358979323846f * 2.0f * (float)i) / (float)N_FFT;
        win[i] = 0.5f - 0.5f * cosf(x);
    }
    fbT.assign((size_t)NBIN_PAD * n_mels, 0.0f);
    const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(8000.0f);
    std::vector<float> fp(n_mels + 2);
    for (int i = 0; i < n_mels + 2; i++) fp[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1));
    for (int m = 0; m < n_mels; m++) {
        float fl = fp[m], fc = fp[m + 1], fr = fp[m + 2];
        float dl = fmaxf(fc - fl, 1e-6f), dr = fmaxf(fr - fc, 1e-6f);
        float enorm = 2.0f / fmaxf(fr - fl, 1e-6f);
        for (int k = 0; k < N_FREQ; k++) {
            float f = (float)k * 8000.0f / (float)(N_FREQ - 1);
            float w = fmaxf(fminf((f - fl) / dl, (fr - f) / dr), 0.0f);
            fbT[(size_t)k * n_mels + m] = w * enorm;
        }
    }
}

void wh_launch_mel_stft(hipStream_t s, const float* pcm, long pcm_stride, const int* n_samples, int n_clips,
                        long max_frames, const double* tw, const float* win, const float* fbT, int n_mels, float* raw,
                        long raw_clip_stride, long raw_row_stride, unsigned* gmax) {
    dim3 grid((unsigned)((max_frames + 15) / 16), (unsigned)n_clips);
    const size_t sm = (size_t)2 * 16 * N_HALF * sizeof(cplx);       // the two FFT buffers: 50 KiB
    wh_ensure_dyn_lds((const void*)k_mel_stft<16>, sm);
    hipLaunchKernelGGL(k_mel_stft<16>, grid, dim3(16 * 32), sm, s, pcm, pcm_stride, n_samples, tw, win, fbT, n_mels, raw,
                       raw_clip_stride, raw_row_stride, gmax);
}

void wh_launch_mel_norm(hipStream_t s, const float* raw, long raw_row_stride, const unsigned* gmax, int n_mels,
                        long n_frames, float* out) {
    long tot = (long)n_mels * n_frames;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mel_norm, dim3(blocks), dim3(256), 0, s, raw, raw_row_stride, gmax, n_mels, n_frames, out);
}

template <typename T>
void wh_launch_mel_tokens(hipStream_t s, const float* src, long src_clip_stride, long src_row_stride,
                          const int* src_index, const int* frame_start, const int* n_frames_src, const unsigned* gmax,
                          int mode, int n_mels, int n_out, T* tok, long tok_clip_stride) {
    dim3 grid((WH_N_FRAMES + 63) / 64, (unsigned)n_out);
    hipLaunchKernelGGL(k_mel_tokens<T>, grid, dim3(256), 0, s, src, src_clip_stride, src_row_stride, src_index,
                       frame_start, n_frames_src, gmax, mode, n_mels, tok, tok_clip_stride);
}
template void wh_launch_mel_tokens<float>(hipStream_t, const float*, long, long, const int*, const int*, const int*,
                                          const unsigned*, int, int, int, float*, long);
template void wh_launch_mel_tokens<bf16>(hipStream_t, const float*, long, long, const int*, const int*, const int*,
                                         const unsigned*, int, int, int, bf16*, long);
