// wh_json.h — minimal JSON reader (config.json, generation_config.json, safetensors headers,
// --discovery-best-json).  Header-only; numbers are kept as double plus the raw text so 64-bit
// offsets survive.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace whjson {

struct Value;
typedef std::shared_ptr<Value> Ptr;

struct Value {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;  // Str payload, or raw text of a Num
    std::vector<Ptr> arr;
    std::vector<std::pair<std::string, Ptr>> obj;  // insertion order kept

    const Value* get(const std::string& k) const {
        for (auto& kv : obj)
            if (kv.first == k) return kv.second.get();
        return nullptr;
    }
    bool is(Kind k) const { return kind == k; }
    int64_t as_i64(int64_t dflt = 0) const {
        if (kind == Num) {
            if (str.find_first_of(".eEN") != std::string::npos) return (int64_t)num;
            return strtoll(str.c_str(), nullptr, 10);
        }
        if (kind == Bool) return b ? 1 : 0;
        return dflt;
    }
};

class Parser {
  public:
    explicit Parser(const std::string& s) : p_(s.c_str()), end_(s.c_str() + s.size()) {}
    Ptr parse(std::string* err) {
        Ptr v = value();
        ws();
        if (!v || p_ != end_) {
            if (err) *err = err_.empty() ? "trailing characters" : err_;
            return nullptr;
        }
        return v;
    }

  private:
    const char *p_, *end_;
    std::string err_;
    void ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) p_++;
    }
    Ptr fail(const char* m) {
        if (err_.empty()) err_ = m;
        return nullptr;
    }
    Ptr value() {
        ws();
        if (p_ >= end_) return fail("unexpected end");
        char c = *p_;
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') {
            auto v = std::make_shared<Value>();
            v->kind = Value::Str;
            if (!string(v->str)) return nullptr;
            return v;
        }
        if (!strncmp(p_, "true", 4) && end_ - p_ >= 4) { p_ += 4; auto v = std::make_shared<Value>(); v->kind = Value::Bool; v->b = true; return v; }
        if (!strncmp(p_, "false", 5) && end_ - p_ >= 5) { p_ += 5; auto v = std::make_shared<Value>(); v->kind = Value::Bool; return v; }
        if (!strncmp(p_, "null", 4) && end_ - p_ >= 4) { p_ += 4; return std::make_shared<Value>(); }
        if (!strncmp(p_, "NaN", 3) && end_ - p_ >= 3) { p_ += 3; auto v = std::make_shared<Value>(); v->kind = Value::Num; v->num = 0.0 / 0.0; v->str = "NaN"; return v; }
        return number();
    }
    Ptr number() {
        const char* s = p_;
        if (p_ < end_ && (*p_ == '-' || *p_ == '+')) p_++;
        while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+')) p_++;
        if (p_ == s) return fail("bad token");
        auto v = std::make_shared<Value>();
        v->kind = Value::Num;
        v->str.assign(s, p_ - s);
        v->num = strtod(v->str.c_str(), nullptr);
        return v;
    }
    bool string(std::string& out) {
        p_++;  // opening quote
        while (p_ < end_ && *p_ != '"') {
            if (*p_ == '\\') {
                p_++;
                if (p_ >= end_) break;
                switch (*p_) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end_ - p_ < 5) { fail("bad \\u"); return false; }
                        unsigned cp = (unsigned)strtoul(std::string(p_ + 1, 4).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp >= 0xD800 && cp <= 0xDBFF && end_ - p_ >= 7 && p_[1] == '\\' && p_[2] == 'u') {
                            unsigned lo = (unsigned)strtoul(std::string(p_ + 3, 4).c_str(), nullptr, 16);
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                            p_ += 6;
                        }
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p_;
                }
                p_++;
            } else {
                out += *p_++;
            }
        }
        if (p_ >= end_) { fail("unterminated string"); return false; }
        p_++;
        return true;
    }
    Ptr array() {
        auto v = std::make_shared<Value>();
        v->kind = Value::Arr;
        p_++;
        ws();
        if (p_ < end_ && *p_ == ']') { p_++; return v; }
        for (;;) {
            Ptr e = value();
            if (!e) return nullptr;
            v->arr.push_back(e);
            ws();
            if (p_ < end_ && *p_ == ',') { p_++; continue; }
            if (p_ < end_ && *p_ == ']') { p_++; return v; }
            return fail("expected , or ]");
        }
    }
    Ptr object() {
        auto v = std::make_shared<Value>();
        v->kind = Value::Obj;
        p_++;
        ws();
        if (p_ < end_ && *p_ == '}') { p_++; return v; }
        for (;;) {
            ws();
            if (p_ >= end_ || *p_ != '"') return fail("expected key");
            std::string k;
            if (!string(k)) return nullptr;
            ws();
            if (p_ >= end_ || *p_ != ':') return fail("expected :");
            p_++;
            Ptr e = value();
            if (!e) return nullptr;
            v->obj.emplace_back(k, e);
            ws();
            if (p_ < end_ && *p_ == ',') { p_++; continue; }
            if (p_ < end_ && *p_ == '}') { p_++; return v; }
            return fail("expected , or }");
        }
    }
};

inline Ptr parse(const std::string& text, std::string* err = nullptr) { return Parser(text).parse(err); }

}  // namespace whjson
