// whisper_bench.cpp — the reference CLI's surface (src/main.rs:23-86, 1065-1271) over
// libwhisper_hip.so.  Same flags and defaults, same CSV / per-file JSON / summary JSON / stdout;
// the ort::Session calls and the Rust log-mel are replaced by the C ABI.  CPU-EP tuning flags are
// accepted and echoed in `config_used`; GPU-side extras live under new keys / new flags.
#include <dirent.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <thread>

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
    std::string precision = "bf16";
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
        else if (k == "--help" || k == "-h") {
            printf("Usage: whisper_bench [--audio-dir DIR] [--model-id ID] [--onnx-dir DIR|synthetic:<preset>:<seed>] [--language en] "
                   "[--task transcribe] [--max-new-tokens 128] [--warmup 0] [--limit-files 0] [--discovery-best-json F] "
                   "[--out-csv F] [--out-json F] [--out-summary-json F] [--intra-op N] [--inter-op N] [--write-txt] "
                   "[--tokenizer-json F] [--timestamps] [--chunk-parallelism N] [--chunk-length-s 30] [--overlap-s 5] "
                   "[--device 0] [--precision bf16|f32|fp8] [--max-batch 16] [--synthetic-clips N] [--seed 1000]\n");
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

// deterministic in-memory clip (SURVEY §8d config 3 shape): three enveloped sinusoids + noise
static std::vector<float> synthetic_clip(uint64_t seed) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    auto u01 = [&]() { st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return (double)(z >> 11) / 9007199254740992.0; };
    double fr[3], ph[3];
    for (int j = 0; j < 3; j++) { fr[j] = 80.0 + 3920.0 * u01(); ph[j] = 2 * M_PI * u01(); }
    std::vector<float> x(WH_CLIP_SAMPLES);
    for (size_t i = 0; i < x.size(); i++) {
        double t = (double)i / 16000.0, env = 0.5 - 0.5 * cos(2 * M_PI * 4.0 * t), s = 0;
        for (int j = 0; j < 3; j++) s += sin(2 * M_PI * fr[j] * t + ph[j]);
        double u1 = std::max(u01(), 1e-300), u2 = u01();
        double v = 0.25 * s * env + 0.02 * sqrt(-2 * log(u1)) * cos(2 * M_PI * u2);
        x[i] = (float)std::max(-1.0, std::min(1.0, v));
    }
    return x;
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

        Tokenizer tok;  // resolve_tokenizer, src/main.rs:574-635 (HF-cache scan not restated)
        if (!trim(a.tokenizer_json).empty()) {
            if (!is_file(trim(a.tokenizer_json))) throw std::runtime_error("tokenizer_json not found: " + trim(a.tokenizer_json));
            load_tokenizer(trim(a.tokenizer_json), tok);
        } else {
            for (const std::string& cand : {a.onnx_dir + "/tokenizer.json", a.model_id + "/tokenizer.json"})
                if (is_file(cand)) { load_tokenizer(cand, tok); break; }
        }
        const bool synthetic_model = a.onnx_dir.rfind("synthetic:", 0) == 0;
        GenCfg gen = load_generation_cfg(a.onnx_dir + "/generation_config.json");
        if (!synthetic_model && !is_dir(a.onnx_dir)) throw std::runtime_error("onnx_dir does not exist or is not a directory: " + a.onnx_dir);

        wh_model* model = nullptr;
        const int prec = a.precision == "f32" ? WH_PREC_F32 : a.precision == "fp8" ? WH_PREC_FP8 : WH_PREC_BF16;
        if (int rc = wh_model_load(a.onnx_dir.c_str(), a.device, prec, &model))
            throw std::runtime_error("Failed to load " + a.onnx_dir + ": libwhisper_hip error " + std::to_string(rc) + ": " + wh_last_error(nullptr));
        wh_ctx* ctx = nullptr;
        if (int rc = wh_ctx_create(model, a.max_batch, &ctx))
            throw std::runtime_error(std::string("wh_ctx_create: ") + std::to_string(rc) + ": " + wh_last_error(nullptr));

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
            for (size_t i = 0; i < a.warmup; i++) { Timing t; transcribe(ctx, a0, a, &tok, gen, t); }
        }
        std::vector<RowOut> rows;
        std::vector<double> e2e, loadl, pre, model_only, dec, rtfl;
        const std::string txt_dir = parent_dir(a.out_csv);
        for (size_t i = 0; i < files.size(); i++) {  // :1164-1213
            const double tl0 = now_s();
            std::vector<float> audio; double dur;
            load(i, audio, dur);
            const double load_s = now_s() - tl0;
            Timing t;
            std::string text = transcribe(ctx, audio, a, &tok, gen, t);
            const double end_to_end = load_s + t.end_to_end_s;
            rows.push_back(make_row(files[i], dur, end_to_end, text));
            loadl.push_back(load_s); pre.push_back(t.preprocess_s); model_only.push_back(t.model_only_s);
            dec.push_back(t.decode_s); e2e.push_back(end_to_end); rtfl.push_back(end_to_end / std::max(dur, 1e-9));
            if (a.write_txt) {
                std::string base = files[i].substr(0, files[i].rfind('.'));
                write_file((txt_dir.empty() ? "." : txt_dir) + "/" + base + ".transcript.txt", trim(text) + "\n");
            }
        }
        write_file(a.out_csv, csv_text(rows));
        write_file(a.out_json, per_file_json(rows));
        JVal summary = JVal::obj();  // :1235-1257 (+ additive keys gpu, rtfx_end_to_end)
        std::vector<double> rtfx;
        for (double r : rtfl) rtfx.push_back(1.0 / std::max(r, 1e-12));
        summary.set("config_used", cfg.json(true)).set("n_files", JVal::integer((long long)rows.size()))
            .set("latency_end_to_end_s", stat_json(stat_block(e2e)))
            .set("breakdown_s", JVal::obj().set("load_s", stat_json(stat_block(loadl))).set("preprocess_s", stat_json(stat_block(pre)))
                                    .set("model_only_s", stat_json(stat_block(model_only))).set("decode_s", stat_json(stat_block(dec))))
            .set("rtf_end_to_end", stat_json(stat_block(rtfl))).set("model_id", JVal::str(a.model_id)).set("onnx_dir", JVal::str(a.onnx_dir))
            .set("language", JVal::str(a.language)).set("task", JVal::str(a.task)).set("max_new_tokens", JVal::integer((long long)a.max_new_tokens))
            .set("tokenizer_json", JVal::str(tok.loaded ? tok.path : "")).set("timestamps", JVal::boolean(a.timestamps))
            .set("notes", JVal::obj().set("longform", JVal::str("Rust approximation: chunked 30s windows with overlap; greedy decode via decoder_with_past"))
                              .set("token_decode", JVal::str(tok.loaded ? "Tokenizer decode (skip_special_tokens=true)" : "Prints token IDs unless you provide tokenizer.json.")))
            .set("rtfx_end_to_end", stat_json(stat_block(rtfx)))
            .set("gpu", JVal::obj().set("backend", JVal::str("libwhisper_hip (gfx950)")).set("device", JVal::integer(a.device))
                            .set("precision", JVal::str(a.precision)).set("max_batch", JVal::integer(a.max_batch)));
        write_file(a.out_summary_json, summary.pretty());
        printf("DONE\n");  // :1261-1268
        printf("Config used:\n%s\n", cfg.json(false).pretty().c_str());
        printf("Per-file CSV: %s\n", a.out_csv.c_str());
        printf("Per-file JSON: %s\n", a.out_json.c_str());
        printf("Summary JSON: %s\n", a.out_summary_json.c_str());
        double p95 = stat_block(e2e).p95;
        if (std::isfinite(p95)) printf("End-to-end p95(s): %.6f\n", p95);
        wh_ctx_free(ctx);
        wh_model_free(model);
    } catch (const std::exception& e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
