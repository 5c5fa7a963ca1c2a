// wh_model.cpp — model lifecycle: replaces build_session x3 (reference src/main.rs:169-202,
// 1099-1108).  Sources of weights:
//   * "synthetic:<preset>:<seed>" — hash-seeded values, bit-identical to
//     whisper-rust-ort_amd/modelspec.py (no checkpoint exists in this pipeline);
//   * a directory with config.json + model.safetensors in the HF Whisper layout (what
//     --onnx-dir would point at once real weights are supplied).
// The f32 master copy is re-laid-out for the gfx950 kernels: q-scale folded into W_q/b_q, Q|K and
// Q|K|V fused, conv weights re-ordered tap-major so Conv1d is a GEMM over overlapping rows, all
// decoder cross-attention K/V projections stacked into one [Ld*2*d][d] matrix, and everything that
// feeds an MFMA converted to the compute dtype.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>

#include "wh_common.h"
#include "wh_internal.h"
#include "wh_json.h"

// ---- error plumbing -------------------------------------------------------------------------
static thread_local std::string g_err;
void wh_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const std::string& wh_global_error() { return g_err; }

bool wh_ensure_dyn_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;  // (device, kernel) -> largest size set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    size_t& cur = done[{dev, kernel}];
    if (bytes <= cur) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        wh_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed on device %d: %s", bytes, dev, hipGetErrorString(e));
        return false;
    }
    cur = bytes;
    return true;
}
int wh_fail_hip(hipError_t e, const char* what, const char* file, int line) {
    wh_set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return WH_ERR_HIP;
}

// ---- presets & canonical tensor table (mirror of modelspec.py) -----------------------------
bool wh_preset_dims(const std::string& name, wh_dims* o) {
    if (name == "nano") *o = wh_dims{80, 128, 2, 2, 2, 256, 1024, 1500, 448};
    else if (name == "micro") *o = wh_dims{80, 256, 4, 2, 3, 1024, 4099, 1500, 448};
    else if (name == "base") *o = wh_dims{80, 512, 8, 6, 6, 2048, 51865, 1500, 448};
    else if (name == "large-v3") *o = wh_dims{128, 1280, 20, 32, 32, 5120, 51866, 1500, 448};
    else return false;
    return true;
}

typedef std::vector<std::pair<std::string, std::vector<int64_t>>> TensorTable;

void wh_tensor_table(const wh_dims& c, TensorTable& t) {
    const int64_t d = c.d_model, F = c.ffn;
    auto attn = [&](const std::string& p) {
        t.push_back({p + ".q_proj.weight", {d, d}});
        t.push_back({p + ".q_proj.bias", {d}});
        t.push_back({p + ".k_proj.weight", {d, d}});
        t.push_back({p + ".v_proj.weight", {d, d}});
        t.push_back({p + ".v_proj.bias", {d}});
        t.push_back({p + ".out_proj.weight", {d, d}});
        t.push_back({p + ".out_proj.bias", {d}});
    };
    auto ln = [&](const std::string& p) {
        t.push_back({p + ".weight", {d}});
        t.push_back({p + ".bias", {d}});
    };
    auto mlp = [&](const std::string& p) {
        t.push_back({p + ".fc1.weight", {F, d}});
        t.push_back({p + ".fc1.bias", {F}});
        t.push_back({p + ".fc2.weight", {d, F}});
        t.push_back({p + ".fc2.bias", {d}});
    };
    const std::string e = "model.encoder", dd = "model.decoder";
    t.push_back({e + ".conv1.weight", {d, c.n_mels, 3}});
    t.push_back({e + ".conv1.bias", {d}});
    t.push_back({e + ".conv2.weight", {d, d, 3}});
    t.push_back({e + ".conv2.bias", {d}});
    t.push_back({e + ".embed_positions.weight", {c.n_audio_ctx, d}});
    for (int i = 0; i < c.enc_layers; i++) {
        std::string p = e + ".layers." + std::to_string(i);
        attn(p + ".self_attn");
        ln(p + ".self_attn_layer_norm");
        mlp(p);
        ln(p + ".final_layer_norm");
    }
    ln(e + ".layer_norm");
    t.push_back({dd + ".embed_tokens.weight", {c.vocab, d}});
    t.push_back({dd + ".embed_positions.weight", {c.n_text_ctx, d}});
    for (int i = 0; i < c.dec_layers; i++) {
        std::string p = dd + ".layers." + std::to_string(i);
        attn(p + ".self_attn");
        ln(p + ".self_attn_layer_norm");
        attn(p + ".encoder_attn");
        ln(p + ".encoder_attn_layer_norm");
        mlp(p);
        ln(p + ".final_layer_norm");
    }
    ln(dd + ".layer_norm");
}

static size_t numel(const std::vector<int64_t>& s) {
    size_t n = 1;
    for (auto v : s) n *= (size_t)v;
    return n;
}
static bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && !s.compare(s.size() - n, n, suf);
}

// ---- hash-seeded values (modelspec.synth_tensor) --------------------------------------------
static uint64_t fnv1a64(const std::string& s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (unsigned char c : s) { h ^= c; h *= 0x100000001B3ull; }
    return h;
}
void wh_synth_weights(const wh_dims& c, uint64_t seed, std::vector<float>& out) {
    TensorTable tt;
    wh_tensor_table(c, tt);
    size_t total = 0;
    for (auto& e : tt) total += numel(e.second);
    out.resize(total);
    const uint64_t GOLD = 0x9E3779B97F4A7C15ull;
    size_t off = 0;
    for (auto& e : tt) {
        const std::string& name = e.first;
        const size_t n = numel(e.second);
        float* dst = out.data() + off;
        off += n;
        if (ends_with(name, "encoder.embed_positions.weight")) {
            // [3P] modeling_whisper.py sinusoids (:55-64): [sin | cos], float64 evaluation
            const int64_t len = e.second[0], ch = e.second[1], half = ch / 2;
            const double inc = log(10000.0) / (double)(half - 1);
            for (int64_t p = 0; p < len; p++)
                for (int64_t i = 0; i < half; i++) {
                    double st = (double)p * exp(-inc * (double)i);
                    dst[p * ch + i] = (float)sin(st);
                    dst[p * ch + half + i] = (float)cos(st);
                }
            continue;
        }
        float offv = 0.0f, amp;
        // modelspec.value_rule: scales under which a random-weight model decodes non-degenerate greedy streams
        if (name == "model.decoder.layer_norm.weight") { offv = 2.0f; amp = 0.2f; }
        else if (ends_with(name, "layer_norm.weight")) { offv = 1.0f; amp = 0.1f; }
        else if (ends_with(name, ".bias")) amp = 0.1f;
        else if (ends_with(name, "embed_tokens.weight")) amp = 0.05f;
        else if (ends_with(name, "decoder.embed_positions.weight")) amp = 0.05f;
        else {
            size_t fan_in = n / (size_t)e.second[0];
            double gain = 1.0;
            if (name.rfind("model.decoder.", 0) == 0 && (ends_with(name, "out_proj.weight") || ends_with(name, "fc2.weight"))) gain = 4.0;
            else if (ends_with(name, "q_proj.weight") || ends_with(name, "k_proj.weight")) gain = 2.5;
            amp = (float)(sqrt(3.0 / (double)fan_in) * gain);
        }
        const float scale = amp / 8388608.0f;
        const uint64_t key = fnv1a64(name) ^ (seed * GOLD);
        auto fill = [=](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                uint64_t z = key + (uint64_t)i * GOLD;
                z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
                z ^= z >> 27; z *= 0x94D049BB133111EBull;
                z ^= z >> 31;
                float v = ((float)(uint32_t)(z >> 40) - 8388608.0f) * scale;
                dst[i] = (offv != 0.0f) ? offv + v : v;
            }
        };
        const size_t nthr = n > (1u << 20) ? 8 : 1;
        if (nthr == 1) fill(0, n);
        else {
            std::vector<std::thread> th;
            const size_t per = (n + nthr - 1) / nthr;
            for (size_t t = 0; t < nthr; t++) th.emplace_back(fill, std::min(n, t * per), std::min(n, (t + 1) * per));
            for (auto& t : th) t.join();
        }
    }
}

// ---- safetensors / config.json --------------------------------------------------------------
static bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}
static float half_to_float(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s << 31;
        else {
            int sh = 0;
            while (!(m & 1024)) { m <<= 1; sh++; }
            m &= 1023;
            u = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) u = (s << 31) | 0x7F800000u | (m << 13);
    else u = (s << 31) | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int wh_load_model_dir(const std::string& dir, wh_dims* dims, std::vector<float>& master, WhPreQuant* pre) {
    std::string txt, err;
    if (!read_file(dir + "/config.json", txt)) {
        wh_set_error("onnx_dir does not hold a config.json: %s", dir.c_str());
        return WH_ERR_IO;
    }
    auto cfg = whjson::parse(txt, &err);
    if (!cfg || !cfg->is(whjson::Value::Obj)) {
        wh_set_error("config.json: %s", err.c_str());
        return WH_ERR_IO;
    }
    auto geti = [&](const char* k, int64_t dflt) {
        auto v = cfg->get(k);
        return v ? v->as_i64(dflt) : dflt;
    };
    dims->n_mels = (int)geti("num_mel_bins", 80);
    dims->d_model = (int)geti("d_model", 0);
    dims->n_heads = (int)geti("encoder_attention_heads", 0);
    dims->enc_layers = (int)geti("encoder_layers", 0);
    dims->dec_layers = (int)geti("decoder_layers", 0);
    dims->ffn = (int)geti("encoder_ffn_dim", 0);
    dims->vocab = (int)geti("vocab_size", 0);
    dims->n_audio_ctx = (int)geti("max_source_positions", 1500);
    dims->n_text_ctx = (int)geti("max_target_positions", 448);
    if (geti("decoder_attention_heads", dims->n_heads) != dims->n_heads || geti("decoder_ffn_dim", dims->ffn) != dims->ffn) {
        wh_set_error("config.json: encoder/decoder head count or ffn width differ (unsupported)");
        return WH_ERR_UNSUPPORTED;
    }
    // safetensors: u64 header length, JSON header, raw little-endian data
    const std::string st = dir + "/model.safetensors";
    FILE* f = fopen(st.c_str(), "rb");
    if (!f) {
        wh_set_error("Failed to load %s", st.c_str());
        return WH_ERR_IO;
    }
    uint64_t hlen = 0;
    if (fread(&hlen, 8, 1, f) != 1 || hlen > (1ull << 30)) {
        fclose(f);
        wh_set_error("%s: bad safetensors header", st.c_str());
        return WH_ERR_IO;
    }
    std::string hdr(hlen, '\0');
    if (fread(&hdr[0], 1, hlen, f) != hlen) {
        fclose(f);
        wh_set_error("%s: truncated header", st.c_str());
        return WH_ERR_IO;
    }
    auto h = whjson::parse(hdr, &err);
    if (!h || !h->is(whjson::Value::Obj)) {
        fclose(f);
        wh_set_error("%s: header JSON: %s", st.c_str(), err.c_str());
        return WH_ERR_IO;
    }
    TensorTable tt;
    wh_tensor_table(*dims, tt);
    size_t total = 0;
    for (auto& e : tt) total += numel(e.second);
    master.resize(total);
    size_t off = 0;
    std::vector<char> buf;
    for (auto& e : tt) {
        const size_t n = numel(e.second);
        const whjson::Value* ent = h->get(e.first);
        if (!ent && e.first.rfind("model.", 0) == 0) ent = h->get(e.first.substr(6));
        if (!ent) {
            fclose(f);
            wh_set_error("%s: missing tensor %s", st.c_str(), e.first.c_str());
            return WH_ERR_IO;
        }
        const whjson::Value* dt = ent->get("dtype");
        const whjson::Value* sh = ent->get("shape");
        const whjson::Value* offs = ent->get("data_offsets");
        if (!dt || !sh || !offs || offs->arr.size() != 2) {
            fclose(f);
            wh_set_error("%s: malformed entry %s", st.c_str(), e.first.c_str());
            return WH_ERR_IO;
        }
        size_t cnt = 1;
        for (auto& s : sh->arr) cnt *= (size_t)s->as_i64();
        if (cnt != n) {
            fclose(f);
            wh_set_error("%s: tensor %s has %zu elements, expected %zu", st.c_str(), e.first.c_str(), cnt, n);
            return WH_ERR_BAD_SHAPE;
        }
        const int64_t b0 = offs->arr[0]->as_i64(), b1 = offs->arr[1]->as_i64();
        const bool is_f8 = dt->str == "F8_E4M3";
        const size_t esz = dt->str == "F32" ? 4 : (dt->str == "F16" || dt->str == "BF16") ? 2 : is_f8 ? 1 : 0;
        if (!esz || (size_t)(b1 - b0) != n * esz) {
            fclose(f);
            wh_set_error("%s: tensor %s dtype %s unsupported or size mismatch", st.c_str(), e.first.c_str(), dt->str.c_str());
            return WH_ERR_UNSUPPORTED;
        }
        buf.resize(n * esz);
        if (fseek(f, (long)(8 + hlen + b0), SEEK_SET) || fread(buf.data(), 1, buf.size(), f) != buf.size()) {
            fclose(f);
            wh_set_error("%s: short read for %s", st.c_str(), e.first.c_str());
            return WH_ERR_IO;
        }
        float* dst = master.data() + off;
        if (is_f8) {
            // e4m3 codes [rows][cols] + "<name>_scale" F32 [rows] (quantize_fp8.py): master gets dequant * scale
            const size_t rows = (size_t)e.second[0], cols = n / rows;
            const whjson::Value* se = h->get(e.first + "_scale");
            if (!se && e.first.rfind("model.", 0) == 0) se = h->get(e.first.substr(6) + "_scale");
            const whjson::Value* sdt = se ? se->get("dtype") : nullptr;
            const whjson::Value* soffs = se ? se->get("data_offsets") : nullptr;
            if (!se || !sdt || sdt->str != "F32" || !soffs || soffs->arr.size() != 2 ||
                (size_t)(soffs->arr[1]->as_i64() - soffs->arr[0]->as_i64()) != rows * 4) {
                fclose(f);
                wh_set_error("%s: F8_E4M3 tensor %s needs an F32 companion %s_scale with one value per row", st.c_str(), e.first.c_str(), e.first.c_str());
                return WH_ERR_IO;
            }
            std::vector<float> sc(rows);
            if (fseek(f, (long)(8 + hlen + soffs->arr[0]->as_i64()), SEEK_SET) || fread(sc.data(), 4, rows, f) != rows) {
                fclose(f);
                wh_set_error("%s: short read for %s_scale", st.c_str(), e.first.c_str());
                return WH_ERR_IO;
            }
            const uint8_t* codes = (const uint8_t*)buf.data();
            for (size_t r = 0; r < rows; r++)
                for (size_t k = 0; k < cols; k++) dst[r * cols + k] = wh_e4m3_to_f32(codes[r * cols + k]) * sc[r];
            if (pre) {
                WhPreQuantEntry& pq = (*pre)[off];
                pq.codes.assign(codes, codes + n);
                pq.scale = sc;
            }
        } else if (esz == 4) memcpy(dst, buf.data(), n * 4);
        else {
            const uint16_t* s16 = (const uint16_t*)buf.data();
            if (dt->str == "BF16")
                for (size_t i = 0; i < n; i++) { uint32_t u = (uint32_t)s16[i] << 16; memcpy(dst + i, &u, 4); }
            else
                for (size_t i = 0; i < n; i++) dst[i] = half_to_float(s16[i]);
        }
        off += n;
    }
    fclose(f);
    return WH_OK;
}

// ---- device layout ----------------------------------------------------------------------------
static inline uint16_t f32_to_bf16(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

static inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- OCP e4m3fn (1-4-3, bias 7, max 448, no infinities): round to nearest even, saturating.  Bit-identical
// twins: modelspec.quantize_e4m3 (numpy), oracle/whisper_oracle.c e4m3_from_f32; the gfx950 conversion
// instructions agree with both on every input (tools/fp8_check.hip).
uint8_t wh_e4m3_from_f32(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    if (x != x) return 0x7F;
    float a = fabsf(x);
    if (a > 448.0f) a = 448.0f;
    if (a >= 0.015625f) {  // normal: RNE at mantissa bit 20
        memcpy(&u, &a, 4);
        u += 0x7FFFFu + ((u >> 20) & 1u);
        uint32_t code = (((u >> 23) - 120u) << 3) | ((u >> 20) & 7u);
        if (code > 0x7Eu) code = 0x7Eu;
        return (uint8_t)(code | sign);
    }
    return (uint8_t)((uint32_t)nearbyint((double)a * 512.0) | sign);  // multiples of 2^-9; 8 = first normal
}
float wh_e4m3_to_f32(uint8_t c) {
    const int e = (c >> 3) & 15, mnt = c & 7;
    float mag = e == 0 ? (float)mnt * 0.001953125f : ldexpf((float)(8 + mnt), e - 10);
    if ((c & 0x7F) == 0x7F) mag = NAN;
    return (c & 0x80) ? -mag : mag;
}

namespace {
// rows of a Linear weight as e4m3 codes + one scale per row: scale = max|w| / 448 (1 for a zero row),
// code = e4m3(w / scale); `rscale` (a power of two: 1 or head_dim^-0.5) only multiplies the scale
struct QRows {
    std::vector<uint8_t> codes;
    std::vector<float> scale;
};
void quantize_rows(const float* W, size_t rows, size_t cols, float rscale, QRows& q, size_t row0, size_t total_rows) {
    if (q.codes.size() != total_rows * cols) { q.codes.assign(total_rows * cols, 0); q.scale.assign(total_rows, 1.0f); }
    for (size_t r = 0; r < rows; r++) {
        float amax = 0.0f;
        for (size_t k = 0; k < cols; k++) amax = std::max(amax, fabsf(W[r * cols + k]));
        const float sc = amax > 0.0f ? amax / 448.0f : 1.0f;
        for (size_t k = 0; k < cols; k++) q.codes[(row0 + r) * cols + k] = wh_e4m3_from_f32(W[r * cols + k] / sc);
        q.scale[row0 + r] = sc * rscale;
    }
}

// f32 -> fp16 limbs exactly as the device does (wh_common.h x3_split: hi = f16(x) RNE, lo = f16(x - hi))
static inline void f32_to_h2(float x, uint16_t* hi, uint16_t* lo) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    memcpy(hi, &h, 2);
    memcpy(lo, &l, 2);
}
static inline float h2_value(float x) {   // the value an h2 element holds for x
    const _Float16 h = (_Float16)x;
    return (float)h + (float)(_Float16)(x - (float)h);
}

struct Stager {
    std::vector<char> host;
    size_t esz;
    // WH_PREC_F16X3, matrices read as `h2` operands (wh_common.h): rows as 128-byte blocks [32 x fp16 hi | 32 x fp16 lo] per 32
    // columns (cols_pad a multiple of 32) — 4 bytes per element like f32, so sizes and offsets are the f32 ones
    bool planar = false;
    explicit Stager(size_t e) : esz(e) {}
    size_t reserve(size_t bytes) {
        size_t off = (host.size() + 255) & ~(size_t)255;
        host.resize(off + bytes, 0);
        return off;
    }
    // compute-dtype matrix from f32 rows: dst[r][c] for r<rows, c<cols_pad (zero beyond cols)
    size_t put_mat(const float* src, size_t rows, size_t cols, size_t cols_pad, float scale = 1.0f) {
        size_t off = reserve(rows * cols_pad * esz);
        for (size_t r = 0; r < rows; r++) put_row(off, r, cols_pad, src + r * cols, cols, scale);
        return off;
    }
    void put_row(size_t off, size_t r, size_t cols_pad, const float* src, size_t cols, float scale) {
        if (planar) {
            uint16_t* d = (uint16_t*)(host.data() + off) + r * cols_pad * 2;
            for (size_t c = 0; c < cols; c++) f32_to_h2(src[c] * scale, d + (c >> 5) * 64 + (c & 31), d + (c >> 5) * 64 + 32 + (c & 31));
        } else if (esz == 4) {
            float* d = (float*)(host.data() + off) + r * cols_pad;
            for (size_t c = 0; c < cols; c++) d[c] = src[c] * scale;
        } else {
            uint16_t* d = (uint16_t*)(host.data() + off) + r * cols_pad;
            for (size_t c = 0; c < cols; c++) d[c] = f32_to_bf16(src[c] * scale);
        }
    }
    size_t put_bytes(const uint8_t* src, size_t n) {
        size_t off = reserve(n);
        memcpy(host.data() + off, src, n);
        return off;
    }
    // e4m3 code VALUES as a bf16 matrix (exact): the encoder-side GEMM keeps its bf16 operand path and
    // multiplies the accumulator by the row scale
    size_t put_codes_bf16(const QRows& q, size_t rows, size_t cols) {
        size_t off = reserve(rows * cols * 2);
        uint16_t* d = (uint16_t*)(host.data() + off);
        for (size_t i = 0; i < rows * cols; i++) d[i] = f32_to_bf16(wh_e4m3_to_f32(q.codes[i]));
        return off;
    }
    size_t put_f32(const float* src, size_t n, float scale = 1.0f) {
        size_t off = reserve(n * 4);
        float* d = (float*)(host.data() + off);
        for (size_t i = 0; i < n; i++) d[i] = src ? src[i] * scale : 0.0f;
        return off;
    }
};
}  // namespace

int wh_model_build(const wh_dims& c, std::vector<float>&& master_in, int device, int precision, wh_model** out, const WhPreQuant* pre) {
    if (precision != WH_PREC_F32 && precision != WH_PREC_BF16 && precision != WH_PREC_FP8 && precision != WH_PREC_F16X3) {
        wh_set_error("unsupported precision %d", precision);
        return WH_ERR_UNSUPPORTED;
    }
    if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model / c.n_heads != WH_HEAD_DIM || c.d_model % 128 || c.ffn % 128 ||
        c.n_mels % 16 || c.d_model > 1280 || c.n_audio_ctx != 1500 || c.n_text_ctx > 512 || c.vocab <= 0) {
        wh_set_error("unsupported model geometry (need head_dim 64, d_model %% 128 == 0, d_model <= 1280, n_mels %% 16 == 0, 1500 audio positions)");
        return WH_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        wh_set_error("no HIP device available: libwhisper_hip has no CPU fallback");
        return WH_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        wh_set_error("device %d out of range (%d devices)", device, ndev);
        return WH_ERR_ARG;
    }
    WH_HIP_CHECK(hipSetDevice(device));
    auto* m = new wh_model();
    m->dims = c;
    m->prec = precision;
    m->device = device;
    m->esz = (precision == WH_PREC_F32 || precision == WH_PREC_F16X3) ? 4 : 2;   // WH_PREC_F16X3: f32 storage, fp16 limbs made at the fragment loads
    m->master = std::move(master_in);
    TensorTable tt;
    wh_tensor_table(c, tt);
    size_t off = 0;
    for (auto& e : tt) {
        m->index[e.first] = {off, numel(e.second)};
        off += numel(e.second);
    }
    if (off != m->master.size()) {
        wh_set_error("weight blob has %zu floats, geometry needs %zu", m->master.size(), off);
        delete m;
        return WH_ERR_BAD_SHAPE;
    }
    auto T = [&](const std::string& n) -> const float* { return m->master.data() + m->index.at(n).first; };
    const size_t d = c.d_model, F = c.ffn, C = c.n_mels;
    const float qs = 1.0f / sqrtf((float)WH_HEAD_DIM);  // 0.125: exact, folded into W_q, b_q
    Stager st(m->esz);
    std::vector<float> tmp;
    const bool f8 = precision == WH_PREC_FP8;
    const size_t NONE = (size_t)-1;
    // rows of a Linear weight as codes + scales: taken from the checkpoint when it came quantised, else computed here
    auto qrows = [&](const float* W, size_t rows, size_t cols, float rscale, QRows& q, size_t row0, size_t total_rows) {
        if (pre) {
            auto it = pre->find((size_t)(W - m->master.data()));
            if (it != pre->end() && it->second.codes.size() == rows * cols) {
                if (q.codes.size() != total_rows * cols) { q.codes.assign(total_rows * cols, 0); q.scale.assign(total_rows, 1.0f); }
                memcpy(q.codes.data() + row0 * cols, it->second.codes.data(), rows * cols);
                for (size_t r = 0; r < rows; r++) q.scale[row0 + r] = it->second.scale[r] * rscale;
                return;
            }
        }
        quantize_rows(W, rows, cols, rscale, q, row0, total_rows);
    };

    // conv weights tap-major: Wr[o][k*Cin + c] = W[o][c][k]
    auto conv_reorder = [&](const float* w, size_t cout, size_t cin, size_t kpad) {
        tmp.assign(cout * kpad, 0.0f);
        for (size_t o = 0; o < cout; o++)
            for (size_t ci = 0; ci < cin; ci++)
                for (size_t k = 0; k < 3; k++) tmp[o * kpad + k * cin + ci] = w[(o * cin + ci) * 3 + k];
        return st.put_mat(tmp.data(), cout, kpad, kpad);
    };
    m->conv1_k = (int)((3 * C + 31) / 32 * 32);
    const std::string e = "model.encoder", dd = "model.decoder";
    // WH_PREC_F16X3: every matrix a matrix-core kernel reads is stored as fp16 limb pairs (`h2`, wh_common.h) — except conv1, whose
    // activation rows (overlapping windows of the token-major log-mel) cannot be h2 blocks: f32 rows, split at the fragment loads
    const bool x3 = precision == WH_PREC_F16X3;
    size_t o_conv1 = conv_reorder(T(e + ".conv1.weight"), d, C, m->conv1_k);
    st.planar = x3;
    size_t o_conv1b = st.put_f32(T(e + ".conv1.bias"), d);
    size_t o_conv2 = conv_reorder(T(e + ".conv2.weight"), d, d, 3 * d);
    size_t o_conv2b = st.put_f32(T(e + ".conv2.bias"), d);
    size_t o_encpos = st.put_f32(T(e + ".embed_positions.weight"), (size_t)c.n_audio_ctx * d);
    struct EncOff { size_t qk, qkb, v, vb, o, ob, f1, f1b, f2, f2b, l1w, l1b, l2w, l2b, qksc, vsc, osc, f1sc, f2sc, qk8, v8, f18, f28; };
    // WH_PREC_FP8, encoder side: code values as a bf16 matrix + the row scales
    // (+ the raw codes for the fp8-MFMA kernel, wh_gemm8_mx.hip, when `raw8` is given)
    auto put_q = [&](const float* W, size_t rows, size_t cols, size_t* sc_off, size_t* raw8 = nullptr) {
        QRows q;
        qrows(W, rows, cols, 1.0f, q, 0, rows);
        *sc_off = st.put_f32(q.scale.data(), rows);
        if (raw8) *raw8 = st.put_bytes(q.codes.data(), q.codes.size());
        return st.put_codes_bf16(q, rows, cols);
    };
    std::vector<EncOff> eo(c.enc_layers);
    for (int i = 0; i < c.enc_layers; i++) {
        std::string p = e + ".layers." + std::to_string(i);
        EncOff& x = eo[i];
        x.qksc = x.vsc = x.osc = x.f1sc = x.f2sc = x.qk8 = x.v8 = x.f18 = x.f28 = NONE;
        if (f8) {
            QRows q;
            qrows(T(p + ".self_attn.q_proj.weight"), d, d, qs, q, 0, 2 * d);
            qrows(T(p + ".self_attn.k_proj.weight"), d, d, 1.0f, q, d, 2 * d);
            x.qk = st.put_codes_bf16(q, 2 * d, d);
            x.qksc = st.put_f32(q.scale.data(), 2 * d);
            x.qk8 = st.put_bytes(q.codes.data(), q.codes.size());
        } else {
            x.qk = st.reserve(2 * d * d * m->esz);
            for (size_t r = 0; r < d; r++) {
                st.put_row(x.qk, r, d, T(p + ".self_attn.q_proj.weight") + r * d, d, qs);
                st.put_row(x.qk, d + r, d, T(p + ".self_attn.k_proj.weight") + r * d, d, 1.0f);
            }
        }
        tmp.assign(2 * d, 0.0f);
        for (size_t r = 0; r < d; r++) tmp[r] = T(p + ".self_attn.q_proj.bias")[r] * qs;
        x.qkb = st.put_f32(tmp.data(), 2 * d);
        x.v = f8 ? put_q(T(p + ".self_attn.v_proj.weight"), d, d, &x.vsc, &x.v8) : st.put_mat(T(p + ".self_attn.v_proj.weight"), d, d, d);
        x.vb = st.put_f32(T(p + ".self_attn.v_proj.bias"), d);
        x.o = f8 ? put_q(T(p + ".self_attn.out_proj.weight"), d, d, &x.osc) : st.put_mat(T(p + ".self_attn.out_proj.weight"), d, d, d);
        x.ob = st.put_f32(T(p + ".self_attn.out_proj.bias"), d);
        x.f1 = f8 ? put_q(T(p + ".fc1.weight"), F, d, &x.f1sc, &x.f18) : st.put_mat(T(p + ".fc1.weight"), F, d, d);
        x.f1b = st.put_f32(T(p + ".fc1.bias"), F);
        x.f2 = f8 ? put_q(T(p + ".fc2.weight"), d, F, &x.f2sc, &x.f28) : st.put_mat(T(p + ".fc2.weight"), d, F, F);
        x.f2b = st.put_f32(T(p + ".fc2.bias"), d);
        x.l1w = st.put_f32(T(p + ".self_attn_layer_norm.weight"), d);
        x.l1b = st.put_f32(T(p + ".self_attn_layer_norm.bias"), d);
        x.l2w = st.put_f32(T(p + ".final_layer_norm.weight"), d);
        x.l2b = st.put_f32(T(p + ".final_layer_norm.bias"), d);
    }
    size_t o_elnw = st.put_f32(T(e + ".layer_norm.weight"), d), o_elnb = st.put_f32(T(e + ".layer_norm.bias"), d);
    size_t o_tok = st.put_mat(T(dd + ".embed_tokens.weight"), c.vocab, d, d);
    size_t o_dpos = st.put_f32(T(dd + ".embed_positions.weight"), (size_t)c.n_text_ctx * d);
    struct DecOff { size_t qkv, qkvb, qkvs, o, ob, cq, cqb, cqs, co, cob, f1, f1b, f1s, f2, f2b, l1w, l1b, l2w, l2b, l3w, l3b,
                           qkvsc, osc, cqsc, cosc, f1sc, f2sc; };
    // WH_PREC_FP8, decoder side: raw e4m3 codes + row scales.  A LayerNorm consumer keeps γ on the activation side, so
    // its constants are s[n] = sum_k γ[k] W[n][k] and c[n] = bias[n] + sum_k β[k] W[n][k] of the DEQUANTISED weights.
    auto put_codes = [&](const QRows& q, size_t* sc_off) {
        *sc_off = st.put_f32(q.scale.data(), q.scale.size());
        return st.put_bytes(q.codes.data(), q.codes.size());
    };
    auto ln_consts_q = [&](const QRows& q, size_t row0, size_t rows, size_t cols, const float* gamma, const float* beta,
                           const float* bias, float rscale, float* s_out, float* c_out) {
        for (size_t r = 0; r < rows; r++) {
            double sacc = 0.0, cacc = 0.0;
            for (size_t k = 0; k < cols; k++) {
                const double w = (double)wh_e4m3_to_f32(q.codes[(row0 + r) * cols + k]);
                sacc += (double)gamma[k] * w;
                cacc += (double)beta[k] * w;
            }
            s_out[r] = (float)(sacc * (double)q.scale[row0 + r]);
            c_out[r] = (float)(cacc * (double)q.scale[row0 + r] + (bias ? (double)bias[r] * rscale : 0.0));
        }
    };
    // LayerNorm folded into the consumer GEMM: rows [row0, row0+rows) of the matrix at `off`
    // become W'[r][k] = rscale * W[r][k] * gamma[k] in the compute dtype; s[r] = sum_k W'[r][k] (of the values
    // as stored), c[r] = sum_k beta[k] * rscale * W[r][k] + rscale * bias[r]
    auto fold_ln = [&](size_t off, size_t row0, const float* W, size_t rows, size_t cols, float rscale, const float* gamma,
                       const float* beta, const float* bias, float* s_out, float* c_out) {
        std::vector<float> row(cols);
        for (size_t r = 0; r < rows; r++) {
            double sacc = 0.0, cacc = bias ? (double)bias[r] * rscale : 0.0;
            for (size_t k = 0; k < cols; k++) {
                const float w = W[r * cols + k] * rscale;
                row[k] = w * gamma[k];
                sacc += (m->esz == 2) ? (double)bf16_to_f32(f32_to_bf16(row[k])) : st.planar ? (double)h2_value(row[k]) : (double)row[k];
                cacc += (double)beta[k] * (double)w;
            }
            st.put_row(off, row0 + r, cols, row.data(), cols, 1.0f);
            s_out[r] = (float)sacc;
            c_out[r] = (float)cacc;
        }
    };
    // WH_PREC_BF16: the encoder's LayerNorms folded into the GEMMs they feed (Q|K, V, fc1, and — the final one — the stacked
    // cross-attention K/V projection); the plain matrices stay for contexts that do not take the fold path
    struct EncFold { size_t qk, qks, qkc, v, vs, vc, f1, f1s, f1c; };
    std::vector<EncFold> ef(c.enc_layers);
    const bool enc_fold = m->prec == WH_PREC_BF16;
    if (enc_fold) {
        for (int i = 0; i < c.enc_layers; i++) {
            std::string p = e + ".layers." + std::to_string(i);
            EncFold& x = ef[i];
            const float *g1 = T(p + ".self_attn_layer_norm.weight"), *b1 = T(p + ".self_attn_layer_norm.bias");
            const float *g2 = T(p + ".final_layer_norm.weight"), *b2 = T(p + ".final_layer_norm.bias");
            std::vector<float> sv(std::max(2 * d, F)), cv(std::max(2 * d, F));
            x.qk = st.reserve(2 * d * d * m->esz);
            fold_ln(x.qk, 0, T(p + ".self_attn.q_proj.weight"), d, d, qs, g1, b1, T(p + ".self_attn.q_proj.bias"), sv.data(), cv.data());
            fold_ln(x.qk, d, T(p + ".self_attn.k_proj.weight"), d, d, 1.0f, g1, b1, nullptr, sv.data() + d, cv.data() + d);
            x.qks = st.put_f32(sv.data(), 2 * d);
            x.qkc = st.put_f32(cv.data(), 2 * d);
            x.v = st.reserve(d * d * m->esz);
            fold_ln(x.v, 0, T(p + ".self_attn.v_proj.weight"), d, d, 1.0f, g1, b1, T(p + ".self_attn.v_proj.bias"), sv.data(), cv.data());
            x.vs = st.put_f32(sv.data(), d);
            x.vc = st.put_f32(cv.data(), d);
            x.f1 = st.reserve(F * d * m->esz);
            fold_ln(x.f1, 0, T(p + ".fc1.weight"), F, d, 1.0f, g2, b2, T(p + ".fc1.bias"), sv.data(), cv.data());
            x.f1s = st.put_f32(sv.data(), F);
            x.f1c = st.put_f32(cv.data(), F);
        }
    }
    std::vector<DecOff> dof(c.dec_layers);
    const size_t Ld = c.dec_layers;
    size_t o_ckv = st.reserve(Ld * 2 * d * d * m->esz);
    tmp.assign(Ld * 2 * d, 0.0f);
    std::vector<float> ckvb(Ld * 2 * d, 0.0f);
    std::vector<float> ckvsc(Ld * 2 * d, 1.0f);
    std::vector<uint8_t> ckv8(f8 ? Ld * 2 * d * d : 0);   // WH_PREC_FP8: the stacked cross K/V projection rows as raw e4m3 codes
    // WH_PREC_FP8 on whisper-base geometry: contexts of 256 clips and more attend over e4m3 encoder states (wh_cross_es8.hip)
    const bool es8 = f8 && wh_cross_es_geometry(c.d_model, c.n_heads, c.n_audio_ctx) && getenv("WH_NO_ES8") == nullptr;
    std::vector<std::vector<float>> deq_k(es8 ? Ld : 0), deq_v(es8 ? Ld : 0);
    for (int i = 0; i < c.dec_layers; i++) {
        std::string p = dd + ".layers." + std::to_string(i);
        DecOff& x = dof[i];
        const float *g1 = T(p + ".self_attn_layer_norm.weight"), *b1 = T(p + ".self_attn_layer_norm.bias");
        std::vector<float> s3(3 * d), c3(3 * d), s1(std::max(d, F)), c1(std::max(d, F));
        x.qkvsc = x.osc = x.cqsc = x.cosc = x.f1sc = x.f2sc = NONE;
        if (f8) {
            const float *g2 = T(p + ".encoder_attn_layer_norm.weight"), *b2 = T(p + ".encoder_attn_layer_norm.bias");
            const float *g3 = T(p + ".final_layer_norm.weight"), *b3 = T(p + ".final_layer_norm.bias");
            QRows q;
            qrows(T(p + ".self_attn.q_proj.weight"), d, d, qs, q, 0, 3 * d);
            qrows(T(p + ".self_attn.k_proj.weight"), d, d, 1.0f, q, d, 3 * d);
            qrows(T(p + ".self_attn.v_proj.weight"), d, d, 1.0f, q, 2 * d, 3 * d);
            ln_consts_q(q, 0, d, d, g1, b1, T(p + ".self_attn.q_proj.bias"), qs, s3.data(), c3.data());
            ln_consts_q(q, d, d, d, g1, b1, nullptr, 1.0f, s3.data() + d, c3.data() + d);
            ln_consts_q(q, 2 * d, d, d, g1, b1, T(p + ".self_attn.v_proj.bias"), 1.0f, s3.data() + 2 * d, c3.data() + 2 * d);
            x.qkv = put_codes(q, &x.qkvsc);
            x.qkvb = st.put_f32(c3.data(), 3 * d);
            x.qkvs = st.put_f32(s3.data(), 3 * d);
            QRows qo; qrows(T(p + ".self_attn.out_proj.weight"), d, d, 1.0f, qo, 0, d);
            x.o = put_codes(qo, &x.osc);
            QRows qc; qrows(T(p + ".encoder_attn.q_proj.weight"), d, d, qs, qc, 0, d);
            ln_consts_q(qc, 0, d, d, g2, b2, T(p + ".encoder_attn.q_proj.bias"), qs, s1.data(), c1.data());
            x.cq = put_codes(qc, &x.cqsc);
            x.cqb = st.put_f32(c1.data(), d);
            x.cqs = st.put_f32(s1.data(), d);
            QRows qco; qrows(T(p + ".encoder_attn.out_proj.weight"), d, d, 1.0f, qco, 0, d);
            x.co = put_codes(qco, &x.cosc);
            QRows qk, qv;
            qrows(T(p + ".encoder_attn.k_proj.weight"), d, d, 1.0f, qk, 0, d);
            qrows(T(p + ".encoder_attn.v_proj.weight"), d, d, 1.0f, qv, 0, d);
            if (es8) {   // the quantised model's own W_k and W_v (code x row scale) for the encoder-state form, below
                deq_k[i].resize(d * d);
                deq_v[i].resize(d * d);
                for (size_t r = 0; r < d; r++)
                    for (size_t k = 0; k < d; k++) {
                        deq_k[i][r * d + k] = wh_e4m3_to_f32(qk.codes[r * d + k]) * qk.scale[r];
                        deq_v[i][r * d + k] = wh_e4m3_to_f32(qv.codes[r * d + k]) * qv.scale[r];
                    }
            }
            for (size_t r = 0; r < d; r++) {
                std::vector<float> rowk(d), rowv(d);
                for (size_t k = 0; k < d; k++) { rowk[k] = wh_e4m3_to_f32(qk.codes[r * d + k]); rowv[k] = wh_e4m3_to_f32(qv.codes[r * d + k]); }
                st.put_row(o_ckv, ((size_t)i * 2 + 0) * d + r, d, rowk.data(), d, 1.0f);
                st.put_row(o_ckv, ((size_t)i * 2 + 1) * d + r, d, rowv.data(), d, 1.0f);
                memcpy(ckv8.data() + (((size_t)i * 2 + 0) * d + r) * d, qk.codes.data() + r * d, d);
                memcpy(ckv8.data() + (((size_t)i * 2 + 1) * d + r) * d, qv.codes.data() + r * d, d);
                ckvsc[((size_t)i * 2 + 0) * d + r] = qk.scale[r];
                ckvsc[((size_t)i * 2 + 1) * d + r] = qv.scale[r];
                ckvb[((size_t)i * 2 + 1) * d + r] = T(p + ".encoder_attn.v_proj.bias")[r];
            }
            QRows q1; qrows(T(p + ".fc1.weight"), F, d, 1.0f, q1, 0, F);
            ln_consts_q(q1, 0, F, d, g3, b3, T(p + ".fc1.bias"), 1.0f, s1.data(), c1.data());
            x.f1 = put_codes(q1, &x.f1sc);
            x.f1b = st.put_f32(c1.data(), F);
            x.f1s = st.put_f32(s1.data(), F);
            QRows q2; qrows(T(p + ".fc2.weight"), d, F, 1.0f, q2, 0, d);
            x.f2 = put_codes(q2, &x.f2sc);
        } else {
        x.qkv = st.reserve(3 * d * d * m->esz);
        fold_ln(x.qkv, 0, T(p + ".self_attn.q_proj.weight"), d, d, qs, g1, b1, T(p + ".self_attn.q_proj.bias"), s3.data(), c3.data());
        fold_ln(x.qkv, d, T(p + ".self_attn.k_proj.weight"), d, d, 1.0f, g1, b1, nullptr, s3.data() + d, c3.data() + d);
        fold_ln(x.qkv, 2 * d, T(p + ".self_attn.v_proj.weight"), d, d, 1.0f, g1, b1, T(p + ".self_attn.v_proj.bias"), s3.data() + 2 * d,
                c3.data() + 2 * d);
        x.qkvb = st.put_f32(c3.data(), 3 * d);
        x.qkvs = st.put_f32(s3.data(), 3 * d);
        x.o = st.put_mat(T(p + ".self_attn.out_proj.weight"), d, d, d);
        x.cq = st.reserve(d * d * m->esz);
        fold_ln(x.cq, 0, T(p + ".encoder_attn.q_proj.weight"), d, d, qs, T(p + ".encoder_attn_layer_norm.weight"),
                T(p + ".encoder_attn_layer_norm.bias"), T(p + ".encoder_attn.q_proj.bias"), s1.data(), c1.data());
        x.cqb = st.put_f32(c1.data(), d);
        x.cqs = st.put_f32(s1.data(), d);
        x.co = st.put_mat(T(p + ".encoder_attn.out_proj.weight"), d, d, d);
        for (size_t r = 0; r < d; r++) {
            st.put_row(o_ckv, ((size_t)i * 2 + 0) * d + r, d, T(p + ".encoder_attn.k_proj.weight") + r * d, d, 1.0f);
            st.put_row(o_ckv, ((size_t)i * 2 + 1) * d + r, d, T(p + ".encoder_attn.v_proj.weight") + r * d, d, 1.0f);
            ckvb[((size_t)i * 2 + 1) * d + r] = T(p + ".encoder_attn.v_proj.bias")[r];
        }
        x.f1 = st.reserve(F * d * m->esz);
        fold_ln(x.f1, 0, T(p + ".fc1.weight"), F, d, 1.0f, T(p + ".final_layer_norm.weight"), T(p + ".final_layer_norm.bias"),
                T(p + ".fc1.bias"), s1.data(), c1.data());
        x.f1b = st.put_f32(c1.data(), F);
        x.f1s = st.put_f32(s1.data(), F);
        x.f2 = st.put_mat(T(p + ".fc2.weight"), d, F, F);
        }
        x.ob = st.put_f32(T(p + ".self_attn.out_proj.bias"), d);
        x.cob = st.put_f32(T(p + ".encoder_attn.out_proj.bias"), d);
        x.f2b = st.put_f32(T(p + ".fc2.bias"), d);
        x.l1w = st.put_f32(T(p + ".self_attn_layer_norm.weight"), d);
        x.l1b = st.put_f32(T(p + ".self_attn_layer_norm.bias"), d);
        x.l2w = st.put_f32(T(p + ".encoder_attn_layer_norm.weight"), d);
        x.l2b = st.put_f32(T(p + ".encoder_attn_layer_norm.bias"), d);
        x.l3w = st.put_f32(T(p + ".final_layer_norm.weight"), d);
        x.l3b = st.put_f32(T(p + ".final_layer_norm.bias"), d);
    }
    // Cross-attention on the encoder states (wh_cross_es.hip; bf16, whisper-base geometry): the K and V projections move out of the
    // per-clip cache to the query side and the context side of the attention kernel.  Per layer and head h (rows 64 h .. 64 h + 63):
    //   qe[h][j] = sum_t Wk[64 h + t][j] q[64 h + t]   (k_dec_qexpand)   -> cqx_w = head h's rows of W_k transposed, [H][d][64] (k_proj has no bias)
    //   out[64 h + t] = sum_j Wv[64 h + t][j] ctx[h][j] + bv[64 h + t]   (grouped decode GEMM) -> the plain W_v rows and b_v of the stacked
    //   cross-K/V projection (cross_kv_w / cross_kv_b), which stay on the device for the contexts that project K and V
    // The query projection (LN2 folded, pre-scaled) and the out-projection are the ones the K / V form uses.
    std::vector<size_t> o_cqx(c.dec_layers, NONE), o_cvx(c.dec_layers, NONE);
    // (WH_PREC_F16X3: the same, with both matrices as fp16 limbs.  WH_PREC_FP8, wh_cross_es8.hip: the states are e4m3 and the two matrices are the
    // quantised model's own W_k / W_v — code x row scale — as bf16: the expansion and the per-head V product run on the bf16 kernels)
    const bool cross_es = (m->prec == WH_PREC_BF16 || x3 || es8) && wh_cross_es_geometry(c.d_model, c.n_heads, c.n_audio_ctx);
    if (cross_es) {
        const size_t H = c.n_heads, HD = WH_HEAD_DIM;
        std::vector<float> wkT(H * d * HD);
        for (int i = 0; i < c.dec_layers; i++) {
            const float* Wk = es8 ? deq_k[i].data() : T(dd + ".layers." + std::to_string(i) + ".encoder_attn.k_proj.weight");
            if (es8) o_cvx[i] = st.put_mat(deq_v[i].data(), d, d, d);
            for (size_t h = 0; h < H; h++)
                for (size_t j = 0; j < d; j++)
                    for (size_t t = 0; t < HD; t++) wkT[(h * d + j) * HD + t] = Wk[(h * HD + t) * d + j];
            o_cqx[i] = st.put_mat(wkT.data(), H * d, HD, HD);
        }
    }
    size_t o_ckvb = st.put_f32(ckvb.data(), ckvb.size());
    size_t o_ckvf = NONE, o_ckvs = NONE, o_ckvc = NONE;
    if (enc_fold) {   // the encoder's final LayerNorm folded into the stacked K/V projection rows (k_proj has no bias)
        o_ckvf = st.reserve(Ld * 2 * d * d * m->esz);
        std::vector<float> sv(Ld * 2 * d), cv(Ld * 2 * d);
        const float *ge = T(e + ".layer_norm.weight"), *be = T(e + ".layer_norm.bias");
        for (size_t i = 0; i < Ld; i++) {
            std::string p = dd + ".layers." + std::to_string(i);
            fold_ln(o_ckvf, (i * 2 + 0) * d, T(p + ".encoder_attn.k_proj.weight"), d, d, 1.0f, ge, be, nullptr, sv.data() + (i * 2 + 0) * d, cv.data() + (i * 2 + 0) * d);
            fold_ln(o_ckvf, (i * 2 + 1) * d, T(p + ".encoder_attn.v_proj.weight"), d, d, 1.0f, ge, be, T(p + ".encoder_attn.v_proj.bias"), sv.data() + (i * 2 + 1) * d,
                    cv.data() + (i * 2 + 1) * d);
        }
        o_ckvs = st.put_f32(sv.data(), sv.size());
        o_ckvc = st.put_f32(cv.data(), cv.size());
    }
    size_t o_ckvsc = f8 ? st.put_f32(ckvsc.data(), ckvsc.size()) : NONE;
    size_t o_ckv8 = f8 ? st.put_bytes(ckv8.data(), ckv8.size()) : NONE;
    size_t o_dlnw = st.put_f32(T(dd + ".layer_norm.weight"), d), o_dlnb = st.put_f32(T(dd + ".layer_norm.bias"), d);
    // LM head = tied embedding with the final LayerNorm folded in (a second copy: the plain one stays the lookup table)
    // (WH_PREC_FP8: γ sits on the activation side, so the lookup table itself is the operand)
    std::vector<float> lms(c.vocab), lmc(c.vocab);
    size_t o_lmw = o_tok;
    if (f8) {
        const float *E = T(dd + ".embed_tokens.weight"), *gf = T(dd + ".layer_norm.weight"), *bfin = T(dd + ".layer_norm.bias");
        for (size_t n = 0; n < (size_t)c.vocab; n++) {
            double sacc = 0.0, cacc = 0.0;
            for (size_t k = 0; k < d; k++) {
                const double w = (double)bf16_to_f32(f32_to_bf16(E[n * d + k]));
                sacc += (double)gf[k] * w;
                cacc += (double)bfin[k] * w;
            }
            lms[n] = (float)sacc;
            lmc[n] = (float)cacc;
        }
    } else {
        o_lmw = st.reserve((size_t)c.vocab * d * m->esz);
        fold_ln(o_lmw, 0, T(dd + ".embed_tokens.weight"), c.vocab, d, 1.0f, T(dd + ".layer_norm.weight"), T(dd + ".layer_norm.bias"), nullptr,
                lms.data(), lmc.data());
    }
    size_t o_lms = st.put_f32(lms.data(), c.vocab), o_lmc = st.put_f32(lmc.data(), c.vocab);
    // log-mel tables
    std::vector<double> tw;
    std::vector<float> win, fbT;
    wh_build_mel_tables(c.n_mels, tw, win, fbT);
    size_t o_tw = st.reserve(tw.size() * 8);
    memcpy(st.host.data() + o_tw, tw.data(), tw.size() * 8);
    size_t o_win = st.put_f32(win.data(), win.size());
    size_t o_fb = st.put_f32(fbT.data(), fbT.size());
    st.reserve(4096);  // slack: conv1's K padding may read a few elements past a matrix

    m->arena_bytes = st.host.size();
    hipError_t he = hipMalloc((void**)&m->arena, m->arena_bytes);
    if (he != hipSuccess) {
        delete m;
        return wh_fail_hip(he, "hipMalloc(weights)", __FILE__, __LINE__);
    }
    he = hipMemcpy(m->arena, st.host.data(), m->arena_bytes, hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        hipFree(m->arena);
        delete m;
        return wh_fail_hip(he, "hipMemcpy(weights)", __FILE__, __LINE__);
    }
    auto P = [&](size_t o) { return (void*)(m->arena + o); };
    auto PF = [&](size_t o) { return (float*)(m->arena + o); };
    m->conv1_w = P(o_conv1); m->conv1_b = PF(o_conv1b);
    m->conv2_w = P(o_conv2); m->conv2_b = PF(o_conv2b);
    m->enc_pos = PF(o_encpos);
    m->enc.resize(c.enc_layers);
    for (int i = 0; i < c.enc_layers; i++) {
        EncOff& x = eo[i];
        m->enc[i] = EncLayerDev{P(x.qk), P(x.v), P(x.o), P(x.f1), P(x.f2), PF(x.qkb), PF(x.vb), PF(x.ob), PF(x.f1b),
                                PF(x.f2b), PF(x.l1w), PF(x.l1b), PF(x.l2w), PF(x.l2b)};
        if (f8) { m->enc[i].qk_w8 = P(x.qk8); m->enc[i].v_w8 = P(x.v8); m->enc[i].fc1_w8 = P(x.f18); m->enc[i].fc2_w8 = P(x.f28); }
        if (f8) { m->enc[i].qk_sc = PF(x.qksc); m->enc[i].v_sc = PF(x.vsc); m->enc[i].o_sc = PF(x.osc); m->enc[i].fc1_sc = PF(x.f1sc); m->enc[i].fc2_sc = PF(x.f2sc); }
        if (enc_fold) {
            const EncFold& y = ef[i];
            m->enc[i].qk_wf = P(y.qk); m->enc[i].qk_s = PF(y.qks); m->enc[i].qk_c = PF(y.qkc);
            m->enc[i].v_wf = P(y.v); m->enc[i].v_s = PF(y.vs); m->enc[i].v_c = PF(y.vc);
            m->enc[i].fc1_wf = P(y.f1); m->enc[i].fc1_s = PF(y.f1s); m->enc[i].fc1_c = PF(y.f1c);
        }
    }
    m->enc_ln_w = PF(o_elnw); m->enc_ln_b = PF(o_elnb);
    m->tok_emb = P(o_tok); m->dec_pos = PF(o_dpos);
    m->dec.resize(c.dec_layers);
    for (int i = 0; i < c.dec_layers; i++) {
        DecOff& x = dof[i];
        m->dec[i] = DecLayerDev{P(x.qkv), P(x.o), P(x.cq), P(x.co), P(x.f1), P(x.f2), PF(x.qkvb), PF(x.ob), PF(x.cqb),
                                PF(x.cob), PF(x.f1b), PF(x.f2b), PF(x.l1w), PF(x.l1b), PF(x.l2w), PF(x.l2b), PF(x.l3w),
                                PF(x.l3b), PF(x.qkvs), PF(x.cqs), PF(x.f1s)};
        if (f8) {
            m->dec[i].qkv_sc = PF(x.qkvsc); m->dec[i].o_sc = PF(x.osc); m->dec[i].cq_sc = PF(x.cqsc);
            m->dec[i].co_sc = PF(x.cosc); m->dec[i].fc1_sc = PF(x.f1sc); m->dec[i].fc2_sc = PF(x.f2sc);
        }
    }
    if (cross_es)
        for (int i = 0; i < c.dec_layers; i++) {
            m->dec[i].cqx_w = P(o_cqx[i]);
            m->dec[i].cv_w = o_cvx[i] != NONE ? (void*)P(o_cvx[i]) : (void*)((char*)P(o_ckv) + ((size_t)i * 2 + 1) * d * d * m->esz);
            m->dec[i].cv_b = PF(o_ckvb) + ((size_t)i * 2 + 1) * d;
        }
    m->cross_es = cross_es;
    m->cross_kv_w = P(o_ckv); m->cross_kv_b = PF(o_ckvb);
    if (f8) { m->cross_kv_sc = PF(o_ckvsc); m->cross_kv_w8 = P(o_ckv8); }
    if (enc_fold) { m->cross_kv_wf = P(o_ckvf); m->cross_kv_s = PF(o_ckvs); m->cross_kv_c = PF(o_ckvc); }
    m->dec_ln_w = PF(o_dlnw); m->dec_ln_b = PF(o_dlnb);
    m->lm_w = P(o_lmw); m->lm_s = PF(o_lms); m->lm_c = PF(o_lmc);
    m->mel_tw = (double*)P(o_tw); m->mel_win = PF(o_win); m->mel_fbT = PF(o_fb);
    *out = m;
    return WH_OK;
}
