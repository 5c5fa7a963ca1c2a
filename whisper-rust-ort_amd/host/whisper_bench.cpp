// whisper_bench.cpp — the reference CLI's surface (src/main.rs:23-86, 1065-1271) over
// libwhisper_hip.so.  Same flags and defaults, same CSV / per-file JSON / summary JSON / stdout;
// the ort::Session calls and the Rust log-mel are replaced by the C ABI.  CPU-EP tuning flags are
// accepted and echoed in `config_used`; GPU-side extras live under new keys / new flags.
#include <dirent.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "../../include/whisper_hip.h"
#include "wh_host.h"

using namespace whhost;

struct Args {
    std::string audio_dir = "audio", model_id = "openai/whisper-base", onnx_dir = "whisper-base-with-past";
    std::string language = "en", task = "transcribe";
    size_t max_new_tokens = 128, warmup = 0, limit_files = 0;
    std::string discovery_best_json, out_csv = "results/benchmarks/inference_per_file.csv",
                                     out_json = "results/benchmarks/inference_per_file.json",
                                     out_summary_json = "results/benchmarks/inference_summary.json";
    long long intra_op = 0, inter_op = 0;
    bool write_txt = false, timestamps = false;
    std::string tokenizer_json;
    size_t chunk_parallelism = 0;
    float chunk_length_s = 30.0f, overlap_s = 5.0f;
    // additive (GPU) flags
    int device = 0, max_batch = 16;
    std::string devices;            // "0-7", "0,2,5": one model per listed device (default: --device)
    int streams_per_gpu = 1;        // contexts (HIP streams) per device, one host thread each
    int load_threads = 0;           // host threads that decode / synthesise audio ahead of the GPU (0 = min(16, cores))
    std::string precision = "bf16";
    bool print_plan = false;   // --print-plan: print the device -> context plan of this command line and exit before any device is touched
    size_t synthetic_clips = 0;
    uint64_t seed = 1000;
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void mkdir_p(const std::string& path) {
    size_t pos = 0;
    while ((pos = path.find('/', pos + 1)) != std::string::npos) mkdir(path.substr(0, pos).c_str(), 0755);
    mkdir(path.c_str(), 0755);
}
static std::string parent_dir(const std::string& p) {
    size_t s = p.rfind('/');
    return s == std::string::npos ? "" : p.substr(0, s);
}
static bool is_file(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
static bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
static void write_file(const std::string& p, const std::string& s) {
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + p);
    fwrite(s.data(), 1, s.size(), f);
    fclose(f);
}

static bool parse_args(int argc, char** argv, Args& a) {
    auto need = [&](int& i, const std::string& key, std::string& val, const std::string& inl) {
        if (!inl.empty() || key.find('=') != std::string::npos) { val = inl; return true; }
        if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", key.c_str()); return false; }
        val = argv[++i];
        return true;
    };
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i], inl;
        size_t eq = k.find('=');
        if (eq != std::string::npos) { inl = k.substr(eq + 1); k = k.substr(0, eq); }
        std::string v;
        if (k == "--write-txt") a.write_txt = true;
        else if (k == "--timestamps") a.timestamps = true;
        else if (k == "--print-plan") a.print_plan = true;
        else if (k == "--help" || k == "-h") {
            printf("Usage: whisper_bench [--audio-dir DIR] [--model-id ID] [--onnx-dir DIR|synthetic:<preset>:<seed>] [--language en] "
                   "[--task transcribe] [--max-new-tokens 128] [--warmup 0] [--limit-files 0] [--discovery-best-json F] "
                   "[--out-csv F] [--out-json F] [--out-summary-json F] [--intra-op N] [--inter-op N] [--write-txt] "
                   "[--tokenizer-json F] [--timestamps] [--chunk-parallelism N] [--chunk-length-s 30] [--overlap-s 5] "
                   "[--device 0] [--devices 0-7] [--streams-per-gpu 1] [--load-threads N] [--precision bf16|f32|fp8|f16x3] [--max-batch 16] "
                   "[--synthetic-clips N] [--seed 1000] [--print-plan]\n");
            exit(0);
        } else {
            if (!need(i, argv[i], v, inl)) return false;
            if (k == "--audio-dir") a.audio_dir = v;
            else if (k == "--model-id") a.model_id = v;
            else if (k == "--onnx-dir") a.onnx_dir = v;
            else if (k == "--language") a.language = v;
            else if (k == "--task") a.task = v;
            else if (k == "--max-new-tokens") a.max_new_tokens = strtoull(v.c_str(), nullptr, 10);
            else if (k == "--warmup") a.warmup = strtoull(v.c_str(), nullptr, 10);
            else if (k == "--limit-files") a.limit_files = strtoull(v.c_str(), nullptr, 10);
            else if (k == "--discovery-best-json") a.discovery_best_json = v;
            else if (k == "--out-csv") a.out_csv = v;
            else if (k == "--out-json") a.out_json = v;
            else if (k == "--out-summary-json") a.out_summary_json = v;
            else if (k == "--intra-op") a.intra_op = atoll(v.c_str());
            else if (k == "--inter-op") a.inter_op = atoll(v.c_str());
            else if (k == "--tokenizer-json") a.tokenizer_json = v;
            else if (k == "--chunk-parallelism") a.chunk_parallelism = strtoull(v.c_str(), nullptr, 10);
            else if (k == "--chunk-length-s") a.chunk_length_s = strtof(v.c_str(), nullptr);
            else if (k == "--overlap-s") a.overlap_s = strtof(v.c_str(), nullptr);
            else if (k == "--device") a.device = atoi(v.c_str());
            else if (k == "--devices") a.devices = v;
            else if (k == "--streams-per-gpu") a.streams_per_gpu = std::max(1, atoi(v.c_str()));
            else if (k == "--load-threads") a.load_threads = atoi(v.c_str());
            else if (k == "--precision") a.precision = v;
            else if (k == "--max-batch") a.max_batch = atoi(v.c_str());
            else if (k == "--synthetic-clips") a.synthetic_clips = strtoull(v.c_str(), nullptr, 10);
            else if (k == "--seed") a.seed = strtoull(v.c_str(), nullptr, 10);
            else { fprintf(stderr, "error: unexpected argument '%s'\n", k.c_str()); return false; }
        }
    }
    return true;
}

static OrtCfg suggested_optimum_cfg() {  // src/main.rs:108-122
    unsigned cpu = std::thread::hardware_concurrency();
    if (!cpu) cpu = 8;
    OrtCfg c;
    c.intra_op = std::min<unsigned>(cpu, 16);
    return c;
}

static OrtCfg load_best_cfg_from_discovery(const std::string& path) {  // src/main.rs:124-167
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot read " + path);
    std::string txt; char b[4096]; size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) txt.append(b, n);
    fclose(f);
    std::string err;
    auto j = whjson::parse(txt, &err);
    if (!j) throw std::runtime_error(path + ": " + err);
    const whjson::Value* best = j->get("best");
    OrtCfg c = suggested_optimum_cfg();
    c.inter_op = 1;
    auto get = [&](const char* k) -> const whjson::Value* { return best && best->is(whjson::Value::Obj) ? best->get(k) : nullptr; };
    auto as_bool = [&](const char* k, bool d) {
        auto v = get(k);
        if (!v) return d;
        if (v->is(whjson::Value::Bool)) return v->b;
        if (v->is(whjson::Value::Num)) return v->as_i64() != 0;
        if (v->is(whjson::Value::Str)) { std::string s = lower(trim(v->str)); return s == "1" || s == "true" || s == "yes" || s == "y" || s == "on"; }
        return d;
    };
    auto as_usize = [&](const char* k, long long d) {
        auto v = get(k);
        if (!v) return d;
        if (v->is(whjson::Value::Num)) return (long long)v->as_i64();
        if (v->is(whjson::Value::Str)) { char* e; long long r = strtoll(v->str.c_str(), &e, 10); return (*e || v->str.empty()) ? d : r; }
        return d;
    };
    auto as_str = [&](const char* k, const std::string& d) { auto v = get(k); return (v && v->is(whjson::Value::Str)) ? v->str : d; };
    c.intra_op = as_usize("intra_op", c.intra_op);
    c.inter_op = as_usize("inter_op", 1);
    c.execution_mode = as_str("execution_mode", "SEQUENTIAL");
    c.graph_opt = as_str("graph_opt", "ENABLE_ALL");
    c.cpu_mem_arena = as_bool("cpu_mem_arena", true);
    c.mem_pattern = as_bool("mem_pattern", true);
    c.allow_spinning = as_bool("allow_spinning", true);
    return c;
}

struct GenCfg { std::vector<int64_t> suppress, begin_suppress; };
static GenCfg load_generation_cfg(const std::string& path) {  // src/main.rs:650-657
    GenCfg g;
    if (!is_file(path)) return g;
    FILE* f = fopen(path.c_str(), "rb");
    std::string txt; char b[4096]; size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) txt.append(b, n);
    fclose(f);
    std::string err;
    auto j = whjson::parse(txt, &err);
    if (!j) throw std::runtime_error(path + ": " + err);
    if (auto v = j->get("suppress_tokens")) for (auto& e : v->arr) g.suppress.push_back(e->as_i64());
    if (auto v = j->get("begin_suppress_tokens")) for (auto& e : v->arr) g.begin_suppress.push_back(e->as_i64());
    return g;
}

// deterministic in-memory clip (SURVEY §8d config 3 shape): three enveloped sinusoids (80-4000 Hz) + noise of standard
// deviation 0.02, clipped to [-1, 1].  Built for speed, because the loader threads have to keep a GPU fed that transcribes
// ~900 clips per second: oscillators and the 4 Hz raised-cosine envelope advance by complex rotation in float (re-seeded
// from sin/cos every 1024 samples so the recurrence cannot drift), the noise is a 4-term Irwin-Hall sum (four 16-bit
// uniforms from one splitmix64 draw, variance-matched to N(0,1)): ~3 ms per 30 s clip on one host core.
static std::vector<float> synthetic_clip(uint64_t seed) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next64 = [&]() { st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; };
    auto u01 = [&]() { return (double)(next64() >> 11) / 9007199254740992.0; };
    double fr[4], ph[4];
    for (int j = 0; j < 3; j++) { fr[j] = 80.0 + 3920.0 * u01(); ph[j] = 2 * M_PI * u01(); }
    fr[3] = 4.0; ph[3] = 0.0;   // envelope 0.5 - 0.5 cos(2 pi 4 t)
    std::vector<float> x(WH_CLIP_SAMPLES);
    float re[4], im[4], cr[4], ci[4];
    for (int j = 0; j < 4; j++) { const double w = 2 * M_PI * fr[j] / 16000.0; cr[j] = (float)cos(w); ci[j] = (float)sin(w); }
    const float nscale = 0.02f * 1.7320508f / 32768.0f;   // sum of 4 U(-1/2,1/2) has variance 1/3
    for (size_t i = 0; i < x.size(); i++) {
        if ((i & 1023) == 0)
            for (int j = 0; j < 4; j++) { const double a = 2 * M_PI * fr[j] * ((double)i / 16000.0) + ph[j]; re[j] = (float)cos(a); im[j] = (float)sin(a); }
        const float s = im[0] + im[1] + im[2], env = 0.5f - 0.5f * re[3];
        const uint64_t r = next64();
        const int sum4 = (int)(r & 0xFFFF) + (int)((r >> 16) & 0xFFFF) + (int)((r >> 32) & 0xFFFF) + (int)(r >> 48) - 2 * 65535;
        const float v = 0.25f * s * env + nscale * (float)sum4 * 0.5f;
        x[i] = std::max(-1.0f, std::min(1.0f, v));
        for (int j = 0; j < 4; j++) { const float nr = re[j] * cr[j] - im[j] * ci[j]; im[j] = re[j] * ci[j] + im[j] * cr[j]; re[j] = nr; }
    }
    return x;
}

// A model id of the form org/name that was downloaded earlier has its files under
// $HF_HOME/hub/models--org--name/snapshots/<revision>/ ($HF_HOME defaults to $HOME/.cache/huggingface): the tokenizer.json of
// the most recently modified snapshot directory that holds one (reference src/main.rs:597-633).  "" when there is none.
static std::string hf_cache_tokenizer(const std::string& model_id) {
    const size_t slash = model_id.find('/');
    if (slash == std::string::npos || slash == 0 || slash + 1 >= model_id.size()) return "";
    const std::string org = model_id.substr(0, slash), name = model_id.substr(slash + 1);
    std::string base;
    if (const char* h = getenv("HF_HOME")) base = h;
    else { const char* home = getenv("HOME"); base = std::string(home ? home : ".") + "/.cache/huggingface"; }
    const std::string snaps = base + "/hub/models--" + org + "--" + name + "/snapshots";
    DIR* d = opendir(snaps.c_str());
    if (!d) return "";
    std::string best;
    bool have = false;
    struct timespec best_m = {0, 0};
    while (struct dirent* e = readdir(d)) {
        const std::string rev = e->d_name;
        if (rev == "." || rev == "..") continue;
        const std::string dir = snaps + "/" + rev, cand = dir + "/tokenizer.json";
        struct stat st_dir, st_tok;
        if (stat(cand.c_str(), &st_tok) != 0 || !S_ISREG(st_tok.st_mode) || stat(dir.c_str(), &st_dir) != 0) continue;
        const bool newer = st_dir.st_mtim.tv_sec > best_m.tv_sec || (st_dir.st_mtim.tv_sec == best_m.tv_sec && st_dir.st_mtim.tv_nsec > best_m.tv_nsec);
        if (!have || newer) { have = true; best_m = st_dir.st_mtim; best = cand; }
    }
    closedir(d);
    return best;
}

static std::vector<int> parse_devices(const std::string& spec, int fallback) {   // "0-7", "0,2,5", "" -> {fallback}
    std::vector<int> out;
    if (trim(spec).empty()) { out.push_back(fallback); return out; }
    size_t pos = 0;
    while (pos <= spec.size()) {
        size_t c = spec.find(',', pos);
        std::string part = trim(spec.substr(pos, c == std::string::npos ? std::string::npos : c - pos));
        if (!part.empty()) {
            size_t dash = part.find('-');
            int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
            if (hi < lo) throw std::runtime_error("bad --devices range: " + part);
            for (int d = lo; d <= hi; d++) out.push_back(d);
        }
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    if (out.empty()) throw std::runtime_error("--devices names no device");
    return out;
}

struct Timing { double preprocess_s = 0, model_only_s = 0, decode_s = 0, end_to_end_s = 0; };

// transcribe_longform_chunked (src/main.rs:834-1008) over the C ABI
static std::string transcribe(wh_ctx* ctx, const std::vector<float>& audio, const Args& a, const Tokenizer* tok,
                              const GenCfg& gen, Timing& t) {
    const double t0 = now_s();
    WhisperSpecial sp = special_tokens(a.language, a.task, tok);
    std::vector<int64_t> prompt = {sp.sot, sp.lang, sp.task};
    if (!a.timestamps) prompt.push_back(sp.no_timestamps);
    wh_decode_params p{};
    p.prompt = prompt.data(); p.n_prompt = prompt.size(); p.max_new_tokens = a.max_new_tokens; p.eot = sp.eot;
    p.suppress = gen.suppress.data(); p.n_suppress = gen.suppress.size();
    p.begin_suppress = gen.begin_suppress.data(); p.n_begin_suppress = gen.begin_suppress.size();
    size_t nch = 0;
    wh_longform_plan(audio.size(), a.chunk_length_s, a.overlap_s, nullptr, 0, &nch);
    const size_t stride = prompt.size() + a.max_new_tokens;
    std::vector<int64_t> toks(std::max<size_t>(1, nch) * stride);
    std::vector<size_t> ntok(std::max<size_t>(1, nch));
    size_t got = 0;
    int rc = wh_transcribe_longform(ctx, audio.data(), audio.size(), a.chunk_length_s, a.overlap_s, &p, toks.data(),
                                    ntok.data(), ntok.size(), &got);
    if (rc) throw std::runtime_error(std::string("libwhisper_hip error ") + std::to_string(rc) + ": " + wh_last_error(ctx));
    wh_timing wt{};
    wh_get_timings(ctx, &wt);
    t.preprocess_s = wt.preprocess_s;
    t.model_only_s = wt.encode_s + wt.decode_s;
    const double td0 = now_s();
    std::vector<std::string> texts;
    for (size_t c = 0; c < got; c++) {  // :926-943
        std::vector<int64_t> g;
        if (ntok[c] > prompt.size()) g.assign(toks.begin() + c * stride + prompt.size(), toks.begin() + c * stride + ntok[c]);
        if (!g.empty() && g.back() == sp.eot) g.pop_back();
        std::string text = decode_tokens(g, tok);
        if (text.empty()) text = "[EMPTY]";
        if (text != "[EMPTY]") texts.push_back(text);
    }
    t.decode_s = now_s() - td0;
    std::string full = stitch_texts(texts);
    t.end_to_end_s = now_s() - t0;
    return full;
}

int main(int argc, char** argv) {
    Args a;
    if (!parse_args(argc, argv, a)) return 2;
    try {
        for (auto* p : {&a.out_csv, &a.out_json, &a.out_summary_json})
            if (!parent_dir(*p).empty()) mkdir_p(parent_dir(*p));
        OrtCfg cfg = a.discovery_best_json.empty() ? suggested_optimum_cfg() : load_best_cfg_from_discovery(a.discovery_best_json);
        if (a.intra_op > 0) cfg.intra_op = a.intra_op;
        if (a.inter_op > 0) cfg.inter_op = a.inter_op;

        Tokenizer tok;  // resolve_tokenizer, src/main.rs:574-635: explicit path, model directories, then the Hugging Face cache
        if (!trim(a.tokenizer_json).empty()) {
            if (!is_file(trim(a.tokenizer_json))) throw std::runtime_error("tokenizer_json not found: " + trim(a.tokenizer_json));
            load_tokenizer(trim(a.tokenizer_json), tok);
        } else {
            std::string found;
            for (const std::string& cand : {a.onnx_dir + "/tokenizer.json", a.model_id + "/tokenizer.json"})
                if (is_file(cand)) { found = cand; break; }
            if (found.empty()) found = hf_cache_tokenizer(a.model_id);
            if (!found.empty()) load_tokenizer(found, tok);
        }
        const bool synthetic_model = a.onnx_dir.rfind("synthetic:", 0) == 0;
        GenCfg gen = load_generation_cfg(a.onnx_dir + "/generation_config.json");
        if (!synthetic_model && !is_dir(a.onnx_dir)) throw std::runtime_error("onnx_dir does not exist or is not a directory: " + a.onnx_dir);

        const int prec = a.precision == "f32" ? WH_PREC_F32 : a.precision == "fp8" ? WH_PREC_FP8 : a.precision == "f16x3" ? WH_PREC_F16X3 : WH_PREC_BF16;
        // One model per device (the reference shares its three `&Session`s across the rayon pool, src/main.rs:890-919),
        // `--streams-per-gpu` contexts per model, one host thread per context; files are independent units
        // (src/main.rs:1164 loops over them serially) and are dealt to whichever context is free.
        const std::vector<int> devices = parse_devices(a.devices, a.device);
        if (a.print_plan) {   // what the command line asks for, before wh_model_load touches a device (CPU test of the --devices surface)
            JVal plan = JVal::obj(false);
            std::string devs = "[", ctxl = "[";
            for (size_t i = 0; i < devices.size(); i++) {
                devs += (i ? ", " : "") + std::to_string(devices[i]);
                for (int st = 0; st < a.streams_per_gpu; st++)
                    ctxl += std::string(ctxl.size() > 1 ? ", " : "") + "{\"context\": " + std::to_string(i * a.streams_per_gpu + st) + ", \"device\": " + std::to_string(devices[i]) +
                            ", \"stream\": " + std::to_string(st) + "}";
            }
            JVal jd; jd.raw = devs + "]";
            JVal jc; jc.raw = ctxl + "]";
            plan.set("devices", jd).set("models", JVal::integer((long long)devices.size())).set("contexts", jc)
                .set("host_threads", JVal::integer((long long)(devices.size() * a.streams_per_gpu))).set("max_batch", JVal::integer(a.max_batch))
                .set("precision", JVal::str(a.precision));
            printf("%s\n", plan.pretty().c_str());
            return 0;
        }
        std::vector<wh_model*> models;
        std::vector<wh_ctx*> ctxs;
        for (int dev : devices) {
            wh_model* m = nullptr;
            if (int rc = wh_model_load(a.onnx_dir.c_str(), dev, prec, &m))
                throw std::runtime_error("Failed to load " + a.onnx_dir + " on device " + std::to_string(dev) + ": libwhisper_hip error " +
                                         std::to_string(rc) + ": " + wh_last_error(nullptr));
            models.push_back(m);
            for (int st = 0; st < a.streams_per_gpu; st++) {
                wh_ctx* c = nullptr;
                if (int rc = wh_ctx_create(m, a.max_batch, &c))
                    throw std::runtime_error(std::string("wh_ctx_create: ") + std::to_string(rc) + ": " + wh_last_error(nullptr));
                ctxs.push_back(c);
            }
        }

        std::vector<std::string> files;
        if (a.synthetic_clips) {
            for (size_t i = 0; i < a.synthetic_clips; i++) { char b[64]; snprintf(b, sizeof b, "clip_%04zu.wav", i); files.push_back(b); }
        } else {
            DIR* d = opendir(a.audio_dir.c_str());
            if (!d) throw std::runtime_error("cannot read audio dir " + a.audio_dir);
            while (dirent* e = readdir(d)) {
                std::string n = e->d_name;
                size_t dot = n.rfind('.');
                if (dot == std::string::npos) continue;
                std::string ext = lower(n.substr(dot + 1));
                if (ext == "wav" || ext == "flac" || ext == "mp3") files.push_back(n);
            }
            closedir(d);
        }
        std::sort(files.begin(), files.end());
        if (a.limit_files > 0 && files.size() > a.limit_files) files.resize(a.limit_files);
        if (files.empty()) throw std::runtime_error("No audio files found in " + a.audio_dir);

        auto load = [&](size_t idx, std::vector<float>& audio, double& dur) {
            if (a.synthetic_clips) { audio = synthetic_clip(a.seed + idx); dur = (double)audio.size() / 16000.0; }
            else load_audio_16k_mono(a.audio_dir + "/" + files[idx], audio, &dur);
        };
        if (a.warmup > 0) {  // :1131-1152
            std::vector<float> a0; double d0;
            load(0, a0, d0);
            for (wh_ctx* c : ctxs)
                for (size_t i = 0; i < a.warmup; i++) { Timing t; transcribe(c, a0, a, &tok, gen, t); }
        }

        // ---- the file loop (:1164-1213) as a pipeline (wh_host.h run_file_pipeline): loader threads -> bounded queue -> one
        // worker per context; one-window files are staged in page-locked buffers ----
        typedef PipeItem Item;
        BufferPool pool([]() -> float* {
                            float* p = nullptr;
                            return hipHostMalloc((void**)&p, (size_t)WH_CLIP_SAMPLES * sizeof(float), hipHostMallocDefault) == hipSuccess ? p : nullptr;
                        },
                        [](float* p) { (void)hipHostFree(p); });
        struct Result { std::string text; double dur = 0, load_s = 0; Timing t; bool ok = false; };
        const size_t nfiles = files.size();
        std::vector<Result> results(nfiles);
        const unsigned hc = std::thread::hardware_concurrency();
        const int n_loaders = a.load_threads > 0 ? a.load_threads : (int)std::min<unsigned>(8, std::max<unsigned>(1, hc / 2));
        WhisperSpecial sp = special_tokens(a.language, a.task, &tok);
        std::vector<int64_t> prompt = {sp.sot, sp.lang, sp.task};
        if (!a.timestamps) prompt.push_back(sp.no_timestamps);
        std::vector<double> busy_s(ctxs.size(), 0.0);
        auto process = [&](size_t wi, std::vector<Item>& batch, std::vector<Item>* next) {
            wh_ctx* ctx = ctxs[wi];
            const size_t stride = prompt.size() + a.max_new_tokens;
            const double tb0 = now_s();
            if (batch.size() == 1 && batch[0].n() > (size_t)WH_CLIP_SAMPLES) {
                // a file longer than one window goes alone through the long-form entry (which batches its windows)
                Result& r = results[batch[0].idx];
                r.text = transcribe(ctx, batch[0].audio, a, &tok, gen, r.t);
                r.dur = batch[0].dur; r.load_s = batch[0].load_s; r.ok = true;
            } else {
                // the per-window body of transcribe_longform_chunked (:870-915) for a batch of one-window files
                std::vector<int64_t> toks(batch.size() * stride);
                std::vector<size_t> ntok(batch.size());
                wh_decode_params p{};
                p.prompt = prompt.data(); p.n_prompt = prompt.size(); p.max_new_tokens = a.max_new_tokens; p.eot = sp.eot;
                p.suppress = gen.suppress.data(); p.n_suppress = gen.suppress.size();
                p.begin_suppress = gen.begin_suppress.data(); p.n_begin_suppress = gen.begin_suppress.size();
                std::vector<wh_clip> clips(batch.size()), nclips(next ? next->size() : 0);
                for (size_t k = 0; k < batch.size(); k++) { clips[k].pcm = batch[k].data(); clips[k].n_samples = batch[k].n(); }
                for (size_t k = 0; k < nclips.size(); k++) { nclips[k].pcm = (*next)[k].data(); nclips[k].n_samples = (*next)[k].n(); }
                const double t0 = now_s();
                // the next batch of this worker, if it is loaded already, is copied to the device beside this batch's work
                int rc = wh_transcribe_batch_next(ctx, clips.data(), clips.size(), nclips.empty() ? nullptr : nclips.data(), nclips.size(), &p, toks.data(), ntok.data());
                if (rc) throw std::runtime_error(std::string("libwhisper_hip error ") + std::to_string(rc) + ": " + wh_last_error(ctx));
                const double batch_s = now_s() - t0;
                wh_timing wt{};
                wh_get_timings(ctx, &wt);
                for (size_t k = 0; k < batch.size(); k++) {   // :926-943
                    Result& r = results[batch[k].idx];
                    const double td0 = now_s();
                    std::vector<int64_t> g;
                    if (ntok[k] > prompt.size()) g.assign(toks.begin() + k * stride + prompt.size(), toks.begin() + k * stride + ntok[k]);
                    if (!g.empty() && g.back() == sp.eot) g.pop_back();
                    std::string text = decode_tokens(g, &tok);
                    if (text.empty()) text = "[EMPTY]";
                    std::vector<std::string> texts;
                    if (text != "[EMPTY]") texts.push_back(text);
                    r.text = stitch_texts(texts);
                    r.t.preprocess_s = wt.preprocess_s;
                    r.t.model_only_s = wt.encode_s + wt.decode_s;
                    r.t.decode_s = now_s() - td0;
                    r.t.end_to_end_s = batch_s + r.t.decode_s;   // every clip of a batch completes with its batch
                    r.dur = batch[k].dur; r.load_s = batch[k].load_s; r.ok = true;
                }
            }
            busy_s[wi] += now_s() - tb0;
        };
        // synthetic clips are one-window files by construction: allocate the page-locked staging pool now, outside the timed loop
        // (5.9 GB at --max-batch 1024; a directory of audio files allocates it when the first one-window file shows up)
        if (a.synthetic_clips) pool.ensure(file_pipeline_pool_size((size_t)a.max_batch, ctxs.size(), n_loaders));
        const double loop0 = now_s();
        const std::string first_error = run_file_pipeline(nfiles, n_loaders, ctxs.size(), (size_t)a.max_batch, (size_t)WH_CLIP_SAMPLES, &pool, load, process);
        const double loop_s = now_s() - loop0;
        if (!first_error.empty()) throw std::runtime_error(first_error);

        std::vector<RowOut> rows;
        std::vector<double> e2e, loadl, pre, model_only, dec, rtfl;
        const std::string txt_dir = parent_dir(a.out_csv);
        double audio_total = 0;
        for (size_t i = 0; i < nfiles; i++) {
            const Result& r = results[i];
            if (!r.ok) throw std::runtime_error("file " + files[i] + " was not processed");
            const double end_to_end = r.load_s + r.t.end_to_end_s;   // :1190
            rows.push_back(make_row(files[i], r.dur, end_to_end, r.text));
            loadl.push_back(r.load_s); pre.push_back(r.t.preprocess_s); model_only.push_back(r.t.model_only_s);
            dec.push_back(r.t.decode_s); e2e.push_back(end_to_end); rtfl.push_back(end_to_end / std::max(r.dur, 1e-9));
            audio_total += r.dur;
            if (a.write_txt) {
                std::string base = files[i].substr(0, files[i].rfind('.'));
                write_file((txt_dir.empty() ? "." : txt_dir) + "/" + base + ".transcript.txt", trim(r.text) + "\n");
            }
        }
        const double busy_max = *std::max_element(busy_s.begin(), busy_s.end());
        write_file(a.out_csv, csv_text(rows));
        write_file(a.out_json, per_file_json(rows));
        std::vector<double> rtfx;
        for (double r : rtfl) rtfx.push_back(1.0 / std::max(r, 1e-12));
        SummaryIn si;   // :1235-1257: the reference's keys from wh_host.h reference_summary (pinned to the reference's archived summary)
        si.cfg = cfg; si.end2end = e2e; si.load = loadl; si.preprocess = pre; si.model_only = model_only; si.decode = dec; si.rtf = rtfl;
        si.n_files = rows.size(); si.model_id = a.model_id; si.onnx_dir = a.onnx_dir; si.language = a.language; si.task = a.task;
        si.tokenizer_json = tok.loaded ? tok.path : ""; si.max_new_tokens = (long long)a.max_new_tokens; si.timestamps = a.timestamps;
        JVal summary = reference_summary(si);   // + additive keys gpu, rtfx_end_to_end
        summary
            .set("rtfx_end_to_end", stat_json(stat_block(rtfx)))
            .set("gpu", JVal::obj().set("backend", JVal::str("libwhisper_hip (gfx950)")).set("device", JVal::integer(devices[0]))
                            .set("devices", JVal::integer((long long)devices.size())).set("streams_per_gpu", JVal::integer(a.streams_per_gpu))
                            .set("precision", JVal::str(a.precision)).set("max_batch", JVal::integer(a.max_batch))
                            .set("load_threads", JVal::integer(n_loaders)).set("audio_s", JVal::num(audio_total))
                            // whole-job throughput: audio seconds per wall second of the file loop (loading overlapped), and per
                            // second the busiest context spent inside the library (what bench.py measures with PCM resident)
                            .set("wall_s", JVal::num(loop_s)).set("throughput_rtfx", JVal::num(audio_total / std::max(loop_s, 1e-12)))
                            .set("gpu_busy_s", JVal::num(busy_max)).set("gpu_throughput_rtfx", JVal::num(audio_total / std::max(busy_max, 1e-12))));
        write_file(a.out_summary_json, summary.pretty());
        printf("DONE\n");  // :1261-1268
        printf("Config used:\n%s\n", cfg.json(false).pretty().c_str());
        printf("Per-file CSV: %s\n", a.out_csv.c_str());
        printf("Per-file JSON: %s\n", a.out_json.c_str());
        printf("Summary JSON: %s\n", a.out_summary_json.c_str());
        double p95 = stat_block(e2e).p95;
        if (std::isfinite(p95)) printf("End-to-end p95(s): %.6f\n", p95);
        for (wh_ctx* c : ctxs) wh_ctx_free(c);
        for (wh_model* m : models) wh_model_free(m);
    } catch (const std::exception& e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
