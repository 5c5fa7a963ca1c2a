// wh_host.h — host-side pieces of the reference's binary that sit either side of the hot path,
// restated in C++ (Rust is not available in this image): statistics + emitters byte-compatible with
// serde_json / the csv crate, WAV reader + linear resampler, prompt ids, token → text fallback and
// byte-level BPE decode, long-form stitcher.  Citations: /root/reference/src/main.rs.
#pragma once
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../csrc/wh_json.h"
#include "wh_unicode_lower.h"

namespace whhost {

// ---------------------------------------------------------------------------------------------
// f64 → text exactly as serde_json (ryu `format_finite`): shortest round-trip digits, decimal
// notation while -5 < kk <= 16 else d.ddde±x; integers keep a trailing ".0"; NaN/inf → null.
// ---------------------------------------------------------------------------------------------
inline std::string fmt_f64(double v) {
    if (!std::isfinite(v)) return "null";
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
        if (strtod(buf, nullptr) == v) break;
    }
    // buf = [-]d[.ddd]e±XX
    std::string s(buf);
    bool neg = s[0] == '-';
    if (neg) s.erase(0, 1);
    size_t epos = s.find('e');
    int exp10 = atoi(s.c_str() + epos + 1);
    std::string digits = s.substr(0, epos);
    digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int len = (int)digits.size();
    const int kk = exp10 + 1;  // 10^(kk-1) <= v < 10^kk
    const int k = kk - len;
    std::string out;
    if (0 <= k && kk <= 16) {  // 1234e7 -> 12340000000.0
        out = digits + std::string((size_t)k, '0') + ".0";
    } else if (0 < kk && kk <= 16) {  // 1234e-2 -> 12.34
        out = digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (-5 < kk && kk <= 0) {  // 1234e-6 -> 0.001234
        out = "0." + std::string((size_t)(-kk), '0') + digits;
    } else if (len == 1) {  // 1e30
        out = digits + "e" + std::to_string(kk - 1);
    } else {  // 1234e30 -> 1.234e33
        out = digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    }
    return neg ? "-" + out : out;
}

inline std::string json_escape(const std::string& s) {  // serde_json string escaping
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
                else o += (char)c;
        }
    }
    return o + "\"";
}

// ---------------------------------------------------------------------------------------------
// statistics — src/main.rs:1021-1048
// ---------------------------------------------------------------------------------------------
inline double percentile(std::vector<double> xs, double p) {  // :1021-1031
    if (xs.empty()) return NAN;
    std::sort(xs.begin(), xs.end());
    double k = ((double)xs.size() - 1.0) * (p / 100.0);
    size_t f = (size_t)std::floor(k), c = (size_t)std::ceil(k);
    if (f == c) return xs[f];
    return xs[f] + (xs[c] - xs[f]) * (k - (double)f);
}

struct StatBlock { double min, median, p90, p95, max, mean; };

inline StatBlock stat_block(const std::vector<double>& xs) {  // :1033-1048 (median = v[len/2])
    std::vector<double> v = xs;
    std::sort(v.begin(), v.end());
    StatBlock s;
    s.min = v.empty() ? NAN : v.front();
    s.max = v.empty() ? NAN : v.back();
    double sum = 0;
    for (double x : v) sum += x;
    s.mean = v.empty() ? NAN : sum / (double)v.size();
    s.median = v.empty() ? NAN : v[v.size() / 2];
    s.p90 = percentile(xs, 90.0);
    s.p95 = percentile(xs, 95.0);
    return s;
}

// A tiny ordered-or-sorted JSON writer: serde_json::json! maps serialise with keys in ALPHABETICAL
// order (no preserve_order feature), derived structs in declaration order.
struct JVal {
    enum Kind { Raw, Obj } kind = Raw;
    std::string raw;                                   // already-encoded scalar
    std::vector<std::pair<std::string, JVal>> fields;  // object
    bool sorted = true;
    static JVal num(double v) { JVal j; j.raw = fmt_f64(v); return j; }
    static JVal integer(long long v) { JVal j; j.raw = std::to_string(v); return j; }
    static JVal boolean(bool b) { JVal j; j.raw = b ? "true" : "false"; return j; }
    static JVal str(const std::string& s) { JVal j; j.raw = json_escape(s); return j; }
    static JVal obj(bool sorted_keys = true) { JVal j; j.kind = Obj; j.sorted = sorted_keys; return j; }
    JVal& set(const std::string& k, JVal v) { fields.emplace_back(k, std::move(v)); return *this; }
    void write(std::string& out, int indent) const {  // serde_json PrettyFormatter, 2 spaces
        if (kind == Raw) { out += raw; return; }
        if (fields.empty()) { out += "{}"; return; }
        std::vector<const std::pair<std::string, JVal>*> order;
        for (auto& f : fields) order.push_back(&f);
        if (sorted) std::stable_sort(order.begin(), order.end(), [](auto a, auto b) { return a->first < b->first; });
        out += "{\n";
        for (size_t i = 0; i < order.size(); i++) {
            out += std::string((size_t)(indent + 2), ' ') + json_escape(order[i]->first) + ": ";
            order[i]->second.write(out, indent + 2);
            out += (i + 1 < order.size()) ? ",\n" : "\n";
        }
        out += std::string((size_t)indent, ' ') + "}";
    }
    std::string pretty() const { std::string o; write(o, 0); return o; }
};

inline JVal stat_json(const StatBlock& s) {
    return JVal::obj().set("min", JVal::num(s.min)).set("median", JVal::num(s.median)).set("p90", JVal::num(s.p90))
        .set("p95", JVal::num(s.p95)).set("max", JVal::num(s.max)).set("mean", JVal::num(s.mean));
}

// ---------------------------------------------------------------------------------------------
// rows + emitters — src/main.rs:1053-1060, 1193-1199, 1215-1232
// ---------------------------------------------------------------------------------------------
struct RowOut { std::string file; double duration_s, end_to_end_s, rtf; std::string text; };

inline double round_to(double v, double scale) { return std::round(v * scale) / scale; }  // f64::round: half away from zero

inline RowOut make_row(const std::string& file, double dur, double e2e, const std::string& text) {  // :1190-1199
    RowOut r;
    r.file = file;
    double rtf = e2e / std::max(dur, 1e-9);
    r.duration_s = round_to(dur, 1000.0);
    r.end_to_end_s = round_to(e2e, 10000.0);
    r.rtf = round_to(rtf, 1000000.0);
    r.text = text;
    return r;
}

inline std::string csv_field(const std::string& f) {  // csv crate, QuoteStyle::Necessary
    bool q = f.empty() ? false : false;
    for (char c : f)
        if (c == '"' || c == ',' || c == '\n' || c == '\r') { q = true; break; }
    if (!q) return f;
    std::string o = "\"";
    for (char c : f) { if (c == '"') o += '"'; o += c; }
    return o + "\"";
}

inline std::string csv_text(const std::vector<RowOut>& rows) {  // :1215-1229
    std::string o = "file,duration_s,end_to_end_s,rtf,text\n";
    char b[64];
    for (auto& r : rows) {
        o += csv_field(r.file) + ",";
        snprintf(b, sizeof b, "%.3f", r.duration_s); o += b; o += ",";
        snprintf(b, sizeof b, "%.4f", r.end_to_end_s); o += b; o += ",";
        snprintf(b, sizeof b, "%.6f", r.rtf); o += b; o += ",";
        o += csv_field(r.text) + "\n";
    }
    return o;
}

inline std::string per_file_json(const std::vector<RowOut>& rows) {  // :1232 to_string_pretty(&rows)
    if (rows.empty()) return "[]";
    std::string o = "[\n";
    for (size_t i = 0; i < rows.size(); i++) {
        JVal r = JVal::obj(false);  // derived struct: declaration order
        r.set("file", JVal::str(rows[i].file)).set("duration_s", JVal::num(rows[i].duration_s))
            .set("end_to_end_s", JVal::num(rows[i].end_to_end_s)).set("rtf", JVal::num(rows[i].rtf))
            .set("text", JVal::str(rows[i].text));
        o += "  ";
        r.write(o, 2);
        o += (i + 1 < rows.size()) ? ",\n" : "\n";
    }
    return o + "]";
}

struct OrtCfg {  // :91-100 — CPU-EP knobs: accepted and echoed only, they have no GPU analogue
    long long intra_op = 1, inter_op = 1;
    std::string execution_mode = "SEQUENTIAL", graph_opt = "ENABLE_ALL";
    bool cpu_mem_arena = true, mem_pattern = true, allow_spinning = true;
    JVal json(bool sorted) const {
        JVal j = JVal::obj(sorted);
        j.set("intra_op", JVal::integer(intra_op)).set("inter_op", JVal::integer(inter_op))
            .set("execution_mode", JVal::str(execution_mode)).set("graph_opt", JVal::str(graph_opt))
            .set("cpu_mem_arena", JVal::boolean(cpu_mem_arena)).set("mem_pattern", JVal::boolean(mem_pattern))
            .set("allow_spinning", JVal::boolean(allow_spinning));
        return j;
    }
};

// The summary object of src/main.rs:1235-1257, from the per-file lists of :1200-1205 — exactly the reference's keys (the
// CLI appends its additive gpu{} / rtfx_end_to_end{} keys afterwards).  tests/test_host_cpu.py rebuilds the reference's own
// archived inference_summary.json from its values through this function and requires the bytes to match.
struct SummaryIn {
    OrtCfg cfg;
    std::vector<double> end2end, load, preprocess, model_only, decode, rtf;
    size_t n_files = 0;
    std::string model_id, onnx_dir, language, task, tokenizer_json;   // tokenizer_json: "" when none was loaded (:1250)
    long long max_new_tokens = 128;
    bool timestamps = false;
};
inline JVal reference_summary(const SummaryIn& in) {
    JVal summary = JVal::obj();
    summary.set("config_used", in.cfg.json(true)).set("n_files", JVal::integer((long long)in.n_files))
        .set("latency_end_to_end_s", stat_json(stat_block(in.end2end)))
        .set("breakdown_s", JVal::obj().set("load_s", stat_json(stat_block(in.load))).set("preprocess_s", stat_json(stat_block(in.preprocess)))
                                .set("model_only_s", stat_json(stat_block(in.model_only))).set("decode_s", stat_json(stat_block(in.decode))))
        .set("rtf_end_to_end", stat_json(stat_block(in.rtf))).set("model_id", JVal::str(in.model_id)).set("onnx_dir", JVal::str(in.onnx_dir))
        .set("language", JVal::str(in.language)).set("task", JVal::str(in.task)).set("max_new_tokens", JVal::integer(in.max_new_tokens))
        .set("tokenizer_json", JVal::str(in.tokenizer_json)).set("timestamps", JVal::boolean(in.timestamps))
        .set("notes", JVal::obj().set("longform", JVal::str("Rust approximation: chunked 30s windows with overlap; greedy decode via decoder_with_past"))
                          .set("token_decode", JVal::str(!in.tokenizer_json.empty() ? "Tokenizer decode (skip_special_tokens=true)" : "Prints token IDs unless you provide tokenizer.json.")));
    return summary;
}

// ---------------------------------------------------------------------------------------------
// audio — src/main.rs:207-316 (WAV container only; FLAC/MP3 are out of scope here)
// ---------------------------------------------------------------------------------------------
inline std::vector<float> resample_linear(const std::vector<float>& x, uint32_t sr_in, uint32_t sr_out) {  // :207-226
    if (sr_in == sr_out) return x;
    double ratio = (double)sr_out / (double)sr_in;
    size_t n_out = (size_t)std::llround((double)x.size() * ratio);
    std::vector<float> y;
    y.reserve(n_out);
    for (size_t i = 0; i < n_out; i++) {
        double t = (double)i / ratio;
        long long i0 = (long long)std::floor(t), i1 = i0 + 1;
        double a = t - (double)i0;
        float s0 = (i0 < 0 || (size_t)i0 >= x.size()) ? 0.0f : x[(size_t)i0];
        float s1 = (i1 < 0 || (size_t)i1 >= x.size()) ? 0.0f : x[(size_t)i1];
        y.push_back((float)(1.0 - a) * s0 + (float)a * s1);
    }
    return y;
}

// Decodes PCM WAV (U8 / S16 / IEEE F32), channel-mean downmix (:266-302), resample to 16 kHz.
inline void load_audio_16k_mono(const std::string& path, std::vector<float>& out, double* dur_s) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to open audio: " + path);
    std::vector<unsigned char> d;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    auto u16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) || memcmp(d.data() + 8, "WAVE", 4))
        throw std::runtime_error("Unsupported container (only RIFF/WAVE is decoded here): " + path);
    uint32_t fmt = 0, ch = 0, sr = 0, bits = 0;
    size_t data_off = 0, data_len = 0, p = 12;
    while (p + 8 <= d.size()) {
        uint32_t len = u32(p + 4);
        if (!memcmp(d.data() + p, "fmt ", 4) && p + 8 + 16 <= d.size()) {
            fmt = u16(p + 8); ch = u16(p + 10); sr = u32(p + 12); bits = u16(p + 22);
            if (fmt == 0xFFFE && len >= 26) fmt = u16(p + 8 + 24);  // WAVE_FORMAT_EXTENSIBLE sub-format
        } else if (!memcmp(d.data() + p, "data", 4)) {
            data_off = p + 8;
            data_len = std::min<size_t>(len, d.size() - data_off);
            break;
        }
        p += 8 + len + (len & 1);
    }
    if (!sr) throw std::runtime_error("Unknown sample rate");
    if (!ch) throw std::runtime_error("Unknown channels");
    std::vector<float> s;
    const size_t bps = bits / 8, frames = (bps && ch) ? data_len / (bps * ch) : 0;
    s.reserve(frames);
    for (size_t i = 0; i < frames; i++) {
        float acc = 0.0f;
        for (uint32_t c = 0; c < ch; c++) {
            size_t o = data_off + (i * ch + c) * bps;
            if (fmt == 1 && bits == 8) acc += ((float)d[o] - 128.0f) / 128.0f;
            else if (fmt == 1 && bits == 16) acc += (float)(int16_t)u16(o) / 32768.0f;
            else if (fmt == 3 && bits == 32) { uint32_t u = u32(o); float v; memcpy(&v, &u, 4); acc += v; }
            else throw std::runtime_error("Unsupported decoded sample format");  // :303
        }
        s.push_back(acc / (float)ch);
    }
    if (sr != 16000) s = resample_linear(s, sr, 16000);
    *dur_s = (double)s.size() / 16000.0;
    out.swap(s);
}

// ---------------------------------------------------------------------------------------------
// tokens — src/main.rs:518-569, 637-657
// ---------------------------------------------------------------------------------------------
struct Tokenizer {
    std::map<std::string, int64_t> special;        // added token content → id
    std::vector<std::string> id_to_tok;            // vocab
    std::vector<bool> is_special;
    bool loaded = false;
    std::string path;
};

inline bool load_tokenizer(const std::string& path, Tokenizer& t) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string txt;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) txt.append(buf, n);
    fclose(f);
    std::string err;
    auto j = whjson::parse(txt, &err);
    if (!j) throw std::runtime_error("Failed to load tokenizer " + path + ": " + err);
    auto grow = [&](size_t id) { if (t.id_to_tok.size() <= id) { t.id_to_tok.resize(id + 1); t.is_special.resize(id + 1, false); } };
    if (auto m = j->get("model"))
        if (auto v = m->get("vocab"))
            for (auto& kv : v->obj) { size_t id = (size_t)kv.second->as_i64(); grow(id); t.id_to_tok[id] = kv.first; }
    if (auto a = j->get("added_tokens"))
        for (auto& e : a->arr) {
            auto c = e->get("content"); auto i = e->get("id"); auto sp = e->get("special");
            if (!c || !i) continue;
            size_t id = (size_t)i->as_i64();
            grow(id);
            t.id_to_tok[id] = c->str;
            t.is_special[id] = sp ? sp->b : true;
            t.special[c->str] = (int64_t)id;
        }
    t.loaded = true;
    t.path = path;
    return true;
}

struct WhisperSpecial { int64_t sot, eot, lang, task, no_timestamps; };

inline WhisperSpecial special_tokens(const std::string& language, const std::string& task, const Tokenizer* tok) {
    if (tok && tok->loaded) {  // :529-541
        auto get = [&](const std::string& s) {
            auto it = tok->special.find(s);
            if (it == tok->special.end()) throw std::runtime_error("Tokenizer missing token: " + s);
            return it->second;
        };
        return {get("<|startoftranscript|>"), get("<|endoftext|>"), get("<|" + language + "|>"), get("<|" + task + "|>"),
                get("<|notimestamps|>")};
    }
    WhisperSpecial s;  // :549-566 hard-coded multilingual ids
    s.sot = 50258; s.eot = 50257;
    s.lang = language == "en" ? 50259 : language == "hi" ? 50276 : 50259;
    s.task = task == "transcribe" ? 50359 : task == "translate" ? 50358 : 50359;
    s.no_timestamps = 50363;
    return s;
}

// GPT-2 byte-level alphabet: unicode code point → byte
inline const std::map<uint32_t, unsigned char>& byte_decoder() {
    static std::map<uint32_t, unsigned char> m;
    if (m.empty()) {
        std::vector<int> bs;
        for (int b = '!'; b <= '~'; b++) bs.push_back(b);
        for (int b = 0xA1; b <= 0xAC; b++) bs.push_back(b);
        for (int b = 0xAE; b <= 0xFF; b++) bs.push_back(b);
        std::vector<int> cs = bs;
        int n = 0;
        for (int b = 0; b < 256; b++)
            if (std::find(bs.begin(), bs.end(), b) == bs.end()) { bs.push_back(b); cs.push_back(256 + n++); }
        for (size_t i = 0; i < bs.size(); i++) m[(uint32_t)cs[i]] = (unsigned char)bs[i];
    }
    return m;
}

inline std::string decode_tokens(const std::vector<int64_t>& tokens, const Tokenizer* tok) {  // :637-648
    if (tok && tok->loaded) {  // byte-level BPE decode, skip_special_tokens = true
        std::string joined;
        for (int64_t t : tokens) {
            if (t < 0 || (size_t)t >= tok->id_to_tok.size() || tok->is_special[(size_t)t]) continue;
            joined += tok->id_to_tok[(size_t)t];
        }
        std::string bytes;
        const auto& bd = byte_decoder();
        for (size_t i = 0; i < joined.size();) {
            unsigned char c = (unsigned char)joined[i];
            uint32_t cp; int len;
            if (c < 0x80) { cp = c; len = 1; }
            else if ((c >> 5) == 6) { cp = c & 31; len = 2; }
            else if ((c >> 4) == 14) { cp = c & 15; len = 3; }
            else { cp = c & 7; len = 4; }
            for (int k = 1; k < len && i + k < joined.size(); k++) cp = (cp << 6) | ((unsigned char)joined[i + k] & 63);
            i += (size_t)len;
            auto it = bd.find(cp);
            if (it != bd.end()) bytes += (char)it->second;
        }
        return bytes;
    }
    std::string o = "[TOKENS:";  // :644-647, first 200 ids
    for (size_t i = 0; i < tokens.size() && i < 200; i++) { if (i) o += " "; o += std::to_string(tokens[i]); }
    return o + "]";
}

// ---------------------------------------------------------------------------------------------
// long-form stitcher — src/main.rs:659-696
// ---------------------------------------------------------------------------------------------
// Rust's str::split_whitespace / trim split on the Unicode White_Space property and str::to_lowercase maps every cased
// letter (src/main.rs:663, 671, 686-688 use all three): the helpers below decode UTF-8 and do the same — the White_Space set
// in full, lowercase through the complete table of simple case mappings (wh_unicode_lower.h, generated from the Unicode
// Character Database by tools/gen_unicode_lower.py), the one multi-character mapping (U+0130 -> "i" + U+0307) and the
// contextual final sigma.
inline size_t utf8_next(const std::string& s, size_t i, uint32_t* cp) {   // decodes one scalar value, returns its byte length (malformed: 1 byte, U+FFFD)
    const unsigned char c = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
    if (c < 0x80) { *cp = c; return 1; }
    if ((c & 0xE0) == 0xC0 && cont(1)) { *cp = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu); return 2; }
    if ((c & 0xF0) == 0xE0 && cont(1) && cont(2)) { *cp = ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu); return 3; }
    if ((c & 0xF8) == 0xF0 && cont(1) && cont(2) && cont(3)) {
        *cp = ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) | ((unsigned char)s[i + 3] & 0x3Fu);
        return 4;
    }
    *cp = 0xFFFD;
    return 1;
}
inline void utf8_put(std::string& o, uint32_t cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
}
inline bool is_unicode_ws(uint32_t c) {   // the White_Space property (char::is_whitespace)
    return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 ||
           c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
inline uint32_t lower_cp(uint32_t c) {   // simple lowercase mapping: the generated table (wh_unicode_lower.h), binary search over its runs
    if (c < 0x80) return (c >= 'A' && c <= 'Z') ? c + 32 : c;
    int lo = 0, hi = kLowerRunCount - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        const LowerRun& r = kLowerRuns[mid];
        if (c < r.first) hi = mid - 1;
        else if (c > r.last) lo = mid + 1;
        else return ((c - r.first) % r.stride) == 0 ? (uint32_t)((int32_t)c + r.delta) : c;
    }
    return c;
}
// cased / case-ignorable, as far as the final-sigma rule of str::to_lowercase needs them: a letter with a case mapping in either
// direction is cased; apostrophes, full stop, colon, soft hyphen, combining marks (U+0300-036F) and modifier letters are ignorable
inline bool is_cased(uint32_t c) {
    if (lower_cp(c) != c) return true;
    if ((c >= 'a' && c <= 'z') || (c >= 0xDF && c <= 0xFF && c != 0xF7) || c == 0xB5 || c == 0xAA || c == 0xBA) return true;
    if ((c >= 0x100 && c <= 0x24F) || (c >= 0x3AC && c <= 0x3CE) || (c >= 0x430 && c <= 0x52F) || (c >= 0x561 && c <= 0x587) ||
        (c >= 0x1E00 && c <= 0x1FFF) || (c >= 0xFF41 && c <= 0xFF5A)) return true;
    return false;
}
inline bool is_case_ignorable(uint32_t c) {
    return c == '\'' || c == '.' || c == ':' || c == '^' || c == '`' || c == 0xA8 || c == 0xAD || c == 0xAF || c == 0xB4 || c == 0xB7 || c == 0xB8 ||
           c == 0x2019 || c == 0x2018 || (c >= 0x2B0 && c <= 0x36F) || (c >= 0x483 && c <= 0x489) || (c >= 0x200B && c <= 0x200F);
}
inline std::vector<std::string> split_ws(const std::string& s) {   // str::split_whitespace
    std::vector<std::string> w;
    std::string cur;
    for (size_t i = 0; i < s.size();) {
        uint32_t cp;
        const size_t n = utf8_next(s, i, &cp);
        if (is_unicode_ws(cp)) { if (!cur.empty()) { w.push_back(cur); cur.clear(); } }
        else cur.append(s, i, n);
        i += n;
    }
    if (!cur.empty()) w.push_back(cur);
    return w;
}
inline std::string lower(const std::string& s) {   // str::to_lowercase
    std::string o;
    for (size_t i = 0; i < s.size();) {
        uint32_t cp;
        const size_t n = utf8_next(s, i, &cp);
        if (cp == 0xFFFD && n == 1 && (unsigned char)s[i] >= 0x80) o += s[i];        // malformed byte: passed through
        else if (cp == 0x130) { o += 'i'; utf8_put(o, 0x307); }
        else if (cp == 0x3A3) {
            // final sigma (the one contextual rule of str::to_lowercase): preceded by a cased letter and not followed by one,
            // case-ignorable characters skipped on both sides -> U+03C2, else U+03C3
            bool before = false, after = false;
            for (size_t q = i; q > 0;) {
                size_t b = q - 1;
                while (b > 0 && ((unsigned char)s[b] & 0xC0) == 0x80) b--;
                uint32_t pc;
                utf8_next(s, b, &pc);
                q = b;
                if (is_case_ignorable(pc)) continue;
                before = is_cased(pc);
                break;
            }
            for (size_t q = i + n; q < s.size();) {
                uint32_t nc;
                const size_t m = utf8_next(s, q, &nc);
                q += m;
                if (is_case_ignorable(nc)) continue;
                after = is_cased(nc);
                break;
            }
            utf8_put(o, (before && !after) ? 0x3C2 : 0x3C3);
        }
        else utf8_put(o, lower_cp(cp));
        i += n;
    }
    return o;
}
inline std::string trim(const std::string& s) {   // str::trim
    size_t a = 0, b = s.size();
    while (a < b) {
        uint32_t cp;
        const size_t n = utf8_next(s, a, &cp);
        if (!is_unicode_ws(cp)) break;
        a += n;
    }
    while (b > a) {   // step back one scalar value
        size_t q = b - 1;
        while (q > a && ((unsigned char)s[q] & 0xC0) == 0x80) q--;
        uint32_t cp;
        utf8_next(s, q, &cp);
        if (!is_unicode_ws(cp)) break;
        b = q;
    }
    return s.substr(a, b - a);
}
inline size_t word_overlap(const std::string& a, const std::string& b, size_t max_words) {  // :686-696
    auto aw = split_ws(a), bw = split_ws(b);
    for (auto& w : aw) w = lower(w);
    for (auto& w : bw) w = lower(w);
    size_t mx = std::min(max_words, std::min(aw.size(), bw.size()));
    for (size_t k = mx; k >= 1; k--) {
        if (std::equal(aw.end() - (long)k, aw.end(), bw.begin())) return k;
        if (k == 1) break;
    }
    return 0;
}
inline std::string stitch_texts(const std::vector<std::string>& chunks) {  // :659-684
    std::string out;
    for (auto& chunk : chunks) {
        std::string t = trim(chunk);
        if (t.empty()) continue;
        if (out.empty()) { out = t; continue; }
        size_t ov = word_overlap(out, t, 16);
        if (ov > 0) {
            auto words = split_ws(t);
            std::string rem;
            for (size_t i = ov; i < words.size(); i++) { if (!rem.empty()) rem += " "; rem += words[i]; }
            if (!rem.empty()) { out += " "; out += rem; }
        } else {
            out += " ";
            out += t;
        }
    }
    return out;
}


// ---------------------------------------------------------------------------------------------
// the file loop (src/main.rs:1164-1213) as a pipeline: loader threads -> bounded, index-ordered queue -> one worker
// per context.  The reference loads and transcribes one file after the other and fails fast with the first error;
// here loading runs ahead of the GPUs, rows stay in file order, and the first error — from a loader or a worker — ends
// every thread: each path that sets the error or makes a thread leave wakes all three wait points (items, space, pool).
// ---------------------------------------------------------------------------------------------
struct PipeItem {
    size_t idx = 0;
    std::vector<float> audio;
    double dur = 0, load_s = 0;
    float* pin = nullptr;   // samples in a pool buffer instead of `audio`
    size_t n_pin = 0;
    size_t n() const { return pin ? n_pin : audio.size(); }
    const float* data() const { return pin ? pin : audio.data(); }
};

// Fixed-size staging buffers from caller-supplied alloc / free (page-locked memory in the CLI).  Allocated on first use
// (only runs that meet a one-window file pay for it) and all or nothing: with a partial pool the loader of the next index
// in line could wait for a buffer while later, parked indices hold every one of them.
class BufferPool {
  public:
    BufferPool(std::function<float*()> alloc, std::function<void(float*)> dealloc) : alloc_(std::move(alloc)), free_(std::move(dealloc)) {}
    ~BufferPool() { for (float* p : all_) free_(p); }   // every buffer the pool allocated, wherever an aborted run left it
    // true if the pool holds `n` buffers (allocating them now if this is the first call); false = use pageable memory
    bool ensure(size_t n) {
        std::lock_guard<std::mutex> lk(m_);
        if (tried_) return total_ > 0;
        tried_ = true;
        for (size_t i = 0; i < n; i++) {
            float* p = alloc_();
            if (!p) {   // not enough page-locked memory: give back what was taken, run without the pool — and say so: throughput changes
                for (float* q : free_list_) free_(q);
                free_list_.clear();
                all_.clear();
                fprintf(stderr, "[wh_host] staging pool: %zu of %zu buffers could be allocated; continuing with pageable memory (slower host-to-device copies)\n", i, n);
                return false;
            }
            free_list_.push_back(p);
            all_.push_back(p);
        }
        total_ = n;
        return total_ > 0;
    }
    float* acquire() {   // nullptr once aborted
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return aborted_ || !free_list_.empty(); });
        if (aborted_) return nullptr;
        float* p = free_list_.back();
        free_list_.pop_back();
        return p;
    }
    void release(float* p) {
        { std::lock_guard<std::mutex> lk(m_); free_list_.push_back(p); }
        cv_.notify_one();
    }
    void abort() {
        { std::lock_guard<std::mutex> lk(m_); aborted_ = true; }
        cv_.notify_all();
    }
    size_t total() const { return total_; }

  private:
    std::function<float*()> alloc_;
    std::function<void(float*)> free_;
    std::vector<float*> free_list_, all_;
    size_t total_ = 0;
    bool tried_ = false, aborted_ = false;
    std::mutex m_;
    std::condition_variable cv_;
};

// look-ahead of the loaders (files loaded but not yet handed to a worker: one batch per worker — a worker waits for a FULL batch, so
// anything less would stall it) / buffers a staging pool needs: the look-ahead + the two batches a worker holds (the one it
// transcribes and the one whose host-to-device copy runs beside it) + one per loader
inline size_t file_pipeline_lookahead(size_t max_batch, size_t n_workers) { return max_batch * n_workers + 4; }
inline size_t file_pipeline_pool_size(size_t max_batch, size_t n_workers, int n_loaders) {
    return file_pipeline_lookahead(max_batch, n_workers) + 2 * max_batch * n_workers + (size_t)n_loaders;
}

// load(idx, audio, dur) fills one file; process(worker, batch, next) transcribes a batch of one-window files (<= max_batch of
// them, only what is already loaded) or ONE multi-window file; `next` (may be null) is the batch of one-window files this worker
// will be given next, already loaded — the place to start its host-to-device copy (wh_transcribe_batch_next); both may throw.
// Returns the first error ("" = none).  `pool` (optional) stages one-window files; a batch's buffers go back to the pool after
// the process() call that transcribed it returns.
inline std::string run_file_pipeline(size_t nfiles, int n_loaders, size_t n_workers, size_t max_batch, size_t window_samples, BufferPool* pool,
                                     const std::function<void(size_t, std::vector<float>&, double&)>& load,
                                     const std::function<void(size_t, std::vector<PipeItem>&, std::vector<PipeItem>*)>& process) {
    std::mutex mu;
    std::condition_variable cv_items, cv_space;
    std::deque<PipeItem> ready;               // loaded files, in index order
    std::map<size_t, PipeItem> parked;        // loaded out of order, waiting for their turn
    size_t next_ready = 0;                    // index the queue is waiting for
    size_t handed = 0;                        // files handed to workers so far (under mu)
    std::atomic<size_t> next_load{0};
    std::string first_error;
    const size_t cap = file_pipeline_lookahead(max_batch, n_workers);
    const size_t pool_size = file_pipeline_pool_size(max_batch, n_workers, n_loaders);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto fail = [&](const std::string& what) {   // call WITHOUT mu held
        {
            std::lock_guard<std::mutex> lk(mu);
            if (first_error.empty()) first_error = what.empty() ? "unknown error" : what;
        }
        cv_items.notify_all();
        cv_space.notify_all();
        if (pool) pool->abort();
    };
    auto loader = [&]() {
        for (;;) {
            const size_t i = next_load.fetch_add(1);
            if (i >= nfiles) return;
            { std::lock_guard<std::mutex> lk(mu); if (!first_error.empty()) return; }
            PipeItem it;
            it.idx = i;
            try {
                const double tl0 = now();
                load(i, it.audio, it.dur);
                it.load_s = now() - tl0;
                // one-window files move into a staging buffer of the pool (page-locked in the CLI: the library's host-to-device
                // copy is then one DMA at link speed; a pageable source goes through the runtime's staging buffers)
                if (pool && it.audio.size() <= window_samples && pool->ensure(pool_size)) {
                    it.pin = pool->acquire();
                    if (!it.pin) return;   // aborted: another thread has failed
                    memcpy(it.pin, it.audio.data(), it.audio.size() * sizeof(float));
                    it.n_pin = it.audio.size();
                    std::vector<float>().swap(it.audio);
                }
            } catch (const std::exception& e) {
                fail(e.what());
                return;
            }
            std::unique_lock<std::mutex> lk(mu);
            cv_space.wait(lk, [&] { return !first_error.empty() || i < handed + cap; });   // bounded look-ahead of the consumers
            if (!first_error.empty()) {
                if (it.pin) pool->release(it.pin);
                return;
            }
            parked.emplace(i, std::move(it));
            while (!parked.empty() && parked.begin()->first == next_ready) {
                ready.push_back(std::move(parked.begin()->second));
                parked.erase(parked.begin());
                next_ready++;
            }
            cv_items.notify_all();
        }
    };
    // take the next batch off the queue (under mu): a full batch, the tail of the input, or a multi-window file alone.
    // `block`: wait for one; else only what is available right now.  Returns false when there is none (end of input, error, or not yet).
    auto take = [&](std::unique_lock<std::mutex>& lk, bool block, bool one_window_only, std::vector<PipeItem>& batch) -> bool {
        auto avail = [&] {
            return !first_error.empty() || handed == nfiles || ready.size() >= max_batch || handed + ready.size() == nfiles ||
                   (!ready.empty() && ready.front().n() > window_samples);
        };
        for (;;) {
            if (block) cv_items.wait(lk, avail);
            else if (!avail()) return false;
            if (!first_error.empty()) return false;
            if (ready.empty()) {
                if (handed == nfiles || !block) return false;
                continue;
            }
            break;
        }
        if (ready.front().n() > window_samples) {
            if (one_window_only) return false;   // a multi-window file is never a "next" batch: it goes alone through the long-form entry
            batch.push_back(std::move(ready.front()));
            ready.pop_front();
        } else {
            while (!ready.empty() && batch.size() < max_batch && ready.front().n() <= window_samples) {
                batch.push_back(std::move(ready.front()));
                ready.pop_front();
            }
        }
        handed += batch.size();
        return true;
    };
    auto give_back = [&](std::vector<PipeItem>& batch) {
        for (PipeItem& it : batch)
            if (it.pin) { pool->release(it.pin); it.pin = nullptr; }
        batch.clear();
    };
    auto worker = [&](size_t wi) {
        std::vector<PipeItem> cur, nxt;
        bool have_next = false;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                if (have_next) { cur = std::move(nxt); nxt.clear(); have_next = false; }
                else if (!take(lk, true, false, cur)) { give_back(cur); return; }
                // one batch of look-ahead for this worker, if it is loaded already (never waited for)
                const bool cur_one_window = cur.size() > 1 || cur[0].n() <= window_samples;
                if (cur_one_window) have_next = take(lk, false, true, nxt);
            }
            cv_space.notify_all();
            cv_items.notify_all();
            try {
                process(wi, cur, have_next ? &nxt : nullptr);
                give_back(cur);
            } catch (const std::exception& e) {
                give_back(cur);
                give_back(nxt);
                fail(e.what());
                return;
            }
        }
    };
    std::vector<std::thread> threads;
    for (int i = 0; i < n_loaders; i++) threads.emplace_back(loader);
    for (size_t wi = 0; wi < n_workers; wi++) threads.emplace_back(worker, wi);
    for (auto& t : threads) t.join();
    return first_error;
}

}  // namespace whhost
