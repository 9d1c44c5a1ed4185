// Minimal JSON DOM reader for Waymo scene files (replaces the reference's use of the absent
// nlohmann/json submodule in src/MapReader.cpp:46-53).  Numbers are parsed with strtod (double)
// and narrowed by the caller, which is what nlohmann's get<float>() does
// (src/json_serialization.hpp:12-16).
#pragma once
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace gd {

struct JValue {
    enum Type : uint8_t { Null, Bool, Num, Str, Arr, Obj };
    Type t = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;

    const JValue *find(const char *key) const {
        if (t != Obj) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const JValue &at(const char *key) const {
        const JValue *v = find(key);
        if (!v) throw std::runtime_error(std::string("scene JSON: missing key '") + key + "'");
        return *v;
    }
    bool contains(const char *key) const { return find(key) != nullptr; }
    size_t size() const { return t == Arr ? arr.size() : (t == Obj ? obj.size() : 0); }
    double number() const {
        if (t == Num) return num;
        if (t == Bool) return b ? 1.0 : 0.0;
        throw std::runtime_error("scene JSON: expected a number");
    }
    float f32() const { return static_cast<float>(number()); }
    int64_t i64() const { return static_cast<int64_t>(number()); }
    bool boolean() const {
        if (t == Bool) return b;
        if (t == Num) return num != 0.0;
        throw std::runtime_error("scene JSON: expected a boolean");
    }
    const std::string &string() const {
        if (t != Str) throw std::runtime_error("scene JSON: expected a string");
        return str;
    }
};

class JParser {
  public:
    JParser(const char *begin, const char *end) : p_(begin), e_(end) {}
    JValue parse() {
        JValue v = value(0);
        ws();
        if (p_ != e_) fail("trailing characters");
        return v;
    }

  private:
    const char *p_, *e_;
    [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("scene JSON: ") + m); }
    void ws() {
        while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    JValue value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p_ >= e_) fail("unexpected end");
        JValue v;
        char c = *p_;
        if (c == '{') {
            ++p_;
            v.t = JValue::Obj;
            ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                if (p_ >= e_ || *p_ != '"') fail("expected object key");
                std::string k = str();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            ++p_;
            v.t = JValue::Arr;
            ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return v; }
            for (;;) {
                v.arr.emplace_back(value(depth + 1));
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.t = JValue::Str;
            v.str = str();
        } else if (c == 't' && e_ - p_ >= 4 && !memcmp(p_, "true", 4)) {
            p_ += 4; v.t = JValue::Bool; v.b = true;
        } else if (c == 'f' && e_ - p_ >= 5 && !memcmp(p_, "false", 5)) {
            p_ += 5; v.t = JValue::Bool; v.b = false;
        } else if (c == 'n' && e_ - p_ >= 4 && !memcmp(p_, "null", 4)) {
            p_ += 4; v.t = JValue::Null;
        } else if (c == 'N' && e_ - p_ >= 3 && !memcmp(p_, "NaN", 3)) {
            p_ += 3; v.t = JValue::Num; v.num = std::strtod("nan", nullptr);
        } else if (c == '-' || (c >= '0' && c <= '9') || c == 'I') {
            char *endp = nullptr;
            v.t = JValue::Num;
            v.num = std::strtod(p_, &endp);  // buffer is NUL-terminated by the loader
            if (endp == p_) fail("bad number");
            p_ = endp;
        } else {
            fail("unexpected character");
        }
        return v;
    }
    std::string str() {
        ++p_;  // opening quote
        std::string out;
        while (p_ < e_ && *p_ != '"') {
            if (*p_ == '\\') {
                ++p_;
                if (p_ >= e_) fail("bad escape");
                switch (*p_) {
                case 'n': out.push_back('\n'); break;
                case 't': out.push_back('\t'); break;
                case 'r': out.push_back('\r'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'u': {
                    if (e_ - p_ < 5) fail("bad \\u escape");
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; i++) {
                        char h = p_[i];
                        cp = cp * 16 + (h >= '0' && h <= '9' ? h - '0' : (h | 32) - 'a' + 10);
                    }
                    p_ += 4;
                    if (cp < 0x80) out.push_back(static_cast<char>(cp));
                    else if (cp < 0x800) { out.push_back(static_cast<char>(0xC0 | (cp >> 6))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
                    else { out.push_back(static_cast<char>(0xE0 | (cp >> 12))); out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: out.push_back(*p_);
                }
                ++p_;
            } else {
                out.push_back(*p_++);
            }
        }
        if (p_ >= e_) fail("unterminated string");
        ++p_;
        return out;
    }
};

}  // namespace gd
