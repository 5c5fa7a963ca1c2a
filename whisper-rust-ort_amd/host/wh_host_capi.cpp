// wh_host_capi.cpp — C entry points over wh_host.h so the host logic (statistics, emitters,
// resampler, stitcher, detokeniser) can be unit-tested from Python without a GPU.
#include <cstring>

#include "wh_host.h"

using namespace whhost;

static size_t put(const std::string& s, char* out, size_t cap) {
    if (out && cap) {
        size_t n = std::min(cap - 1, s.size());
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return s.size();
}

extern "C" {

size_t whh_fmt_f64(double v, char* out, size_t cap) { return put(fmt_f64(v), out, cap); }
double whh_percentile(const double* xs, size_t n, double p) { return percentile(std::vector<double>(xs, xs + n), p); }
void whh_stat_block(const double* xs, size_t n, double* out6) {
    StatBlock s = stat_block(std::vector<double>(xs, xs + n));
    out6[0] = s.min; out6[1] = s.median; out6[2] = s.p90; out6[3] = s.p95; out6[4] = s.max; out6[5] = s.mean;
}
size_t whh_stat_json(const double* xs, size_t n, char* out, size_t cap) {
    return put(stat_json(stat_block(std::vector<double>(xs, xs + n))).pretty(), out, cap);
}
// rows: files/texts as NUL-separated lists
static std::vector<RowOut> rows_from(const char* files, const char* texts, const double* dur, const double* e2e, size_t n) {
    std::vector<RowOut> rows;
    for (size_t i = 0; i < n; i++) {
        std::string f(files), t(texts);
        files += f.size() + 1;
        texts += t.size() + 1;
        rows.push_back(make_row(f, dur[i], e2e[i], t));
    }
    return rows;
}
size_t whh_csv(const char* files, const char* texts, const double* dur, const double* e2e, size_t n, char* out, size_t cap) {
    return put(csv_text(rows_from(files, texts, dur, e2e, n)), out, cap);
}
size_t whh_per_file_json(const char* files, const char* texts, const double* dur, const double* e2e, size_t n, char* out, size_t cap) {
    return put(per_file_json(rows_from(files, texts, dur, e2e, n)), out, cap);
}
// the reference's summary object (src/main.rs:1235-1257) from per-file lists: lists = [end2end | load | preprocess | model_only | decode | rtf],
// n values each; strs = NUL-separated model_id, onnx_dir, language, task, tokenizer_json, execution_mode, graph_opt;
// ints = intra_op, inter_op, max_new_tokens; flags bit 0 timestamps, 1 cpu_mem_arena, 2 mem_pattern, 3 allow_spinning
size_t whh_summary_json(const double* lists, size_t n, const char* strs, const long long* ints, unsigned flags, char* out, size_t cap) {
    SummaryIn in;
    auto take = [&](int k) { return std::vector<double>(lists + (size_t)k * n, lists + (size_t)(k + 1) * n); };
    in.end2end = take(0); in.load = take(1); in.preprocess = take(2); in.model_only = take(3); in.decode = take(4); in.rtf = take(5);
    in.n_files = n;
    std::string* dst[7] = {&in.model_id, &in.onnx_dir, &in.language, &in.task, &in.tokenizer_json, &in.cfg.execution_mode, &in.cfg.graph_opt};
    for (auto* d : dst) { *d = std::string(strs); strs += d->size() + 1; }
    in.cfg.intra_op = ints[0]; in.cfg.inter_op = ints[1]; in.max_new_tokens = ints[2];
    in.timestamps = flags & 1; in.cfg.cpu_mem_arena = flags & 2; in.cfg.mem_pattern = flags & 4; in.cfg.allow_spinning = flags & 8;
    return put(reference_summary(in).pretty(), out, cap);
}
size_t whh_lower(const char* s, char* out, size_t cap) { return put(lower(s), out, cap); }
size_t whh_trim(const char* s, char* out, size_t cap) { return put(trim(s), out, cap); }
size_t whh_resample_linear(const float* x, size_t n, unsigned sr_in, unsigned sr_out, float* out, size_t cap) {
    std::vector<float> y = resample_linear(std::vector<float>(x, x + n), sr_in, sr_out);
    if (out) memcpy(out, y.data(), std::min(cap, y.size()) * sizeof(float));
    return y.size();
}
size_t whh_stitch(const char* chunks, size_t n, char* out, size_t cap) {
    std::vector<std::string> v;
    for (size_t i = 0; i < n; i++) { std::string c(chunks); chunks += c.size() + 1; v.push_back(c); }
    return put(stitch_texts(v), out, cap);
}
size_t whh_word_overlap(const char* a, const char* b, size_t max_words) { return word_overlap(a, b, max_words); }
size_t whh_decode_tokens(const long long* toks, size_t n, const char* tokenizer_json, char* out, size_t cap) {
    Tokenizer t;
    std::vector<int64_t> v(toks, toks + n);
    if (tokenizer_json && *tokenizer_json) load_tokenizer(tokenizer_json, t);
    return put(decode_tokens(v, t.loaded ? &t : nullptr), out, cap);
}
int whh_special_tokens(const char* language, const char* task, const char* tokenizer_json, long long* out5) {
    try {
        Tokenizer t;
        if (tokenizer_json && *tokenizer_json) load_tokenizer(tokenizer_json, t);
        WhisperSpecial s = special_tokens(language, task, t.loaded ? &t : nullptr);
        out5[0] = s.sot; out5[1] = s.eot; out5[2] = s.lang; out5[3] = s.task; out5[4] = s.no_timestamps;
        return 0;
    } catch (...) { return 1; }
}
int whh_load_wav(const char* path, float* out, size_t cap, size_t* n, double* dur) {
    try {
        std::vector<float> a;
        load_audio_16k_mono(path, a, dur);
        *n = a.size();
        if (out) memcpy(out, a.data(), std::min(cap, a.size()) * sizeof(float));
        return 0;
    } catch (...) { return 1; }
}
// The CLI's loader / worker pipeline (run_file_pipeline) with stand-in load and process callbacks: file `bad_load` fails to
// load, the batch holding file `bad_process` fails in the worker (either may be >= nfiles: none), `pool_buffers` staging
// buffers can be allocated before the allocator runs dry (0 = no pool).  Returns 0 and the number of processed files, or 1
// and the error text; a deadlock would simply never return — the test runs it under a timeout.
int whh_pipeline_selftest(size_t nfiles, int n_loaders, size_t n_workers, size_t max_batch, size_t bad_load, size_t bad_process,
                          size_t pool_buffers, size_t long_every, size_t* n_processed, char* err, size_t cap) {
    const size_t window = 64;
    std::atomic<size_t> allocated{0}, done{0};
    std::vector<int> seen(nfiles, 0);
    std::mutex sm;
    BufferPool pool([&]() -> float* { return allocated.fetch_add(1) < pool_buffers ? (float*)malloc(window * sizeof(float)) : nullptr; },
                    [](float* p) { free(p); });
    auto load = [&](size_t i, std::vector<float>& audio, double& dur) {
        if (i == bad_load) throw std::runtime_error("cannot decode file " + std::to_string(i));
        const bool is_long = long_every && (i % long_every) == long_every - 1;
        audio.assign(is_long ? 3 * window : window - (i % 5), (float)i);
        dur = (double)audio.size();
        std::this_thread::sleep_for(std::chrono::microseconds(200 + 37 * (i % 7)));
    };
    std::atomic<size_t> with_next{0};
    std::vector<size_t> expect_first(n_workers, (size_t)-1);   // per worker: first index of the batch announced as `next`
    auto process = [&](size_t wi, std::vector<PipeItem>& batch, std::vector<PipeItem>* next) {
        if (batch.empty() || batch.size() > max_batch) throw std::runtime_error("bad batch size");
        if (expect_first[wi] != (size_t)-1 && batch[0].idx != expect_first[wi]) throw std::runtime_error("the announced next batch was not the next batch");
        expect_first[wi] = (size_t)-1;
        if (next) {
            if (next->empty() || next->size() > max_batch) throw std::runtime_error("bad next batch size");
            for (auto& it : *next)
                if (it.n() > window) throw std::runtime_error("a multi-window file was announced as a next batch");
            if ((*next)[0].data()[0] != (float)(*next)[0].idx) throw std::runtime_error("next payload mismatch");
            expect_first[wi] = (*next)[0].idx;
            with_next++;
        }
        if (batch.size() > 1)
            for (auto& it : batch)
                if (it.n() > window) throw std::runtime_error("a multi-window file must go alone");
        for (size_t k = 0; k < batch.size(); k++) {
            if (k && batch[k].idx != batch[k - 1].idx + 1) throw std::runtime_error("batch not in file order");
            if (batch[k].idx == bad_process) throw std::runtime_error("transcribe failed for file " + std::to_string(batch[k].idx));
            if (batch[k].data()[0] != (float)batch[k].idx) throw std::runtime_error("payload mismatch");
            std::lock_guard<std::mutex> lk(sm);
            seen[batch[k].idx]++;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(500));
        done += batch.size();
    };
    const std::string e = run_file_pipeline(nfiles, n_loaders, n_workers, max_batch, window, pool_buffers ? &pool : nullptr, load, process);
    if (n_processed) *n_processed = done.load();
    put(e, err, cap);
    if (!e.empty()) return 1;
    for (size_t i = 0; i < nfiles; i++)
        if (seen[i] != 1) { put("file " + std::to_string(i) + " processed " + std::to_string(seen[i]) + " times", err, cap); return 2; }
    return 0;
}
}
