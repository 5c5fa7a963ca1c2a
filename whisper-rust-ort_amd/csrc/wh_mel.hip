// wh_mel.hip — fused STFT + power + mel filterbank + log10 (+ global max) for gfx950.
//
// Replaces whisper_log_mel_80 (reference src/main.rs:407-509) and its rustfft 400-point FFT
// (src/main.rs:440-441,473).  One workgroup = 16 consecutive frames of one clip:
//   1. the 2800 samples those frames touch are staged once in LDS, reflect padding applied by
//      index arithmetic (src/main.rs:419-435) — no padded copy of the clip exists;
//   2. the real DFT is two small GEMMs on the f64 matrix cores (v_mfma_f64_16x16x4_f64):
//        re[f][k] = sum_{n=0..200} xe[f][n] cos(2 pi n k/400),  xe[n] = xw[n] + xw[400-n]
//        im[f][k] = sum_{n=1..199} xo[f][n] sin(2 pi n k/400),  xo[n] = xw[n] - xw[400-n]
//      (xw = sample * periodic-Hann in f32 exactly as src/main.rs:463-470; the window is
//      symmetric, so the even/odd fold halves the contraction).  Twiddles come from one 400-entry
//      f64 cosine table in LDS.  f64 keeps the spectrum's error far below the f32 FFT's, so the
//      8-decade dynamic range of the log-mel is safe;
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
constexpr int FR_BLK = 16;     // frames per workgroup
constexpr int SPAN = (FR_BLK - 1) * HOP + N_FFT;  // 2800 samples

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

__global__ __launch_bounds__(256) void k_mel_stft(const float* __restrict__ pcm, long pcm_stride,
                                                  const int* __restrict__ n_samples, const double* __restrict__ tw_g,
                                                  const float* __restrict__ win_g, const float* __restrict__ fbT,
                                                  int n_mels, float* __restrict__ raw, long raw_clip_stride,
                                                  long raw_row_stride, unsigned* __restrict__ gmax) {
    __shared__ __attribute__((aligned(16))) double tw[N_FFT];
    __shared__ __attribute__((aligned(16))) float smp[SPAN];
    __shared__ __attribute__((aligned(16))) float win[N_FFT];
    __shared__ __attribute__((aligned(16))) float pw[FR_BLK][NBIN_PAD];

    const int clip = blockIdx.y;
    const long n = n_samples[clip];
    const long n_frames = (n / HOP) > 1 ? (n / HOP) : 1;  // src/main.rs:444-452
    const long f0 = (long)blockIdx.x * FR_BLK;
    if (f0 >= n_frames) return;
    const float* cp = pcm + (long)clip * pcm_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < N_FFT; i += 256) {
        tw[i] = tw_g[i];
        win[i] = win_g[i];
    }
    {   // all of a thread's samples in flight before the LDS stores (SPAN / 256 = 11 loads)
        constexpr int NS = (SPAN + 255) / 256;
        float sv[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) sv[i] = padded_sample(cp, n, f0 * HOP + min(tid + i * 256, SPAN - 1));
#pragma unroll
        for (int i = 0; i < NS; i++)
            if (tid + i * 256 < SPAN) smp[tid + i * 256] = sv[i];
    }
    __syncthreads();

    // ---- DFT on the f64 matrix cores -----------------------------------------------------------
    // A[i = frame (lane&15)][k = n],  B[k = n][j = bin (lane&15)],  n = 4*step + (lane>>4)
    // D (f64 layout): column j = lane&15, rows i = (lane>>4) + 4*r.
    const int fl = lane & 15, g = lane >> 4;
    f64x4 acc_re[4], acc_im[4];
    int bin[4], ic[4], is[4], inc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        acc_re[t] = f64x4{0, 0, 0, 0};
        acc_im[t] = f64x4{0, 0, 0, 0};
        int b = 16 * (wave + 4 * t) + fl;
        bin[t] = b > 200 ? 200 : b;
        ic[t] = (g * bin[t]) % N_FFT;        // (n*bin) mod 400 at step 0 (n = g)
        is[t] = (ic[t] + 300) % N_FFT;       // sin(x) = cos(x - pi/2): table index -100
        inc[t] = (4 * bin[t]) % N_FFT;
    }
    const int ntile = (wave == 0) ? 4 : 3;  // 13 bin tiles over 4 waves
    const float* sf = smp + fl * HOP;
    for (int step = 0; step < 51; step++) {
        const int nn = 4 * step + g;  // 0..203
        double xe = 0.0, xo = 0.0;
        if (nn <= 200) {
            float a = sf[nn] * win[nn];
            if (nn == 0 || nn == 200) {
                xe = (double)a;
            } else {
                float b = sf[N_FFT - nn] * win[N_FFT - nn];
                xe = (double)a + (double)b;
                xo = (double)a - (double)b;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < ntile) {
                double c = tw[ic[t]], s = tw[is[t]];
                acc_re[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xe, c, acc_re[t], 0, 0, 0);
                acc_im[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xo, s, acc_im[t], 0, 0, 0);
                ic[t] += inc[t]; if (ic[t] >= N_FFT) ic[t] -= N_FFT;
                is[t] += inc[t]; if (is[t] >= N_FFT) is[t] -= N_FFT;
            }
        }
    }
    // ---- power spectrum to LDS -----------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < ntile) {
            int b = 16 * (wave + 4 * t) + fl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float re = (float)acc_re[t][r], im = (float)acc_im[t][r];
                float p = re * re + im * im;
                pw[g + 4 * r][b] = (b <= 200) ? p : 0.0f;
            }
        }
    }
    __syncthreads();

    // ---- mel filterbank on the exact-f32 matrix cores ------------------------------------------
    // A[i = frame][k = bin] = pw, B[k = bin][j = mel] = fbT[bin][mel]; D: col j = mel (lane&15),
    // rows i = frame 4*(lane>>4)+r  → each lane owns 4 consecutive frames of one mel row.
    float lmax = -INFINITY;
    const int n_mt = n_mels >> 4;
    for (int mt = wave; mt < n_mt; mt += 4) {
        f32x4 acc = {0, 0, 0, 0};
        const int mel = 16 * mt + fl;
#pragma unroll 13
        for (int step = 0; step < NBIN_PAD / 4; step++) {
            const int k = 4 * step + g;
            float a = pw[fl][k];
            float b = fbT[(long)k * n_mels + mel];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        float* out = raw + (long)clip * raw_clip_stride + (long)mel * raw_row_stride + f0 + 4 * g;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = log10f(fmaxf(acc[r], 1e-10f));  // src/main.rs:489,496
            if (f0 + 4 * g + r < n_frames) lmax = fmaxf(lmax, v[r]);
        }
        if (f0 + 4 * g + 3 < n_frames) {
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
    dim3 grid((unsigned)((max_frames + FR_BLK - 1) / FR_BLK), (unsigned)n_clips);
    hipLaunchKernelGGL(k_mel_stft, grid, dim3(256), 0, s, pcm, pcm_stride, n_samples, tw, win, fbT, n_mels, raw,
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
