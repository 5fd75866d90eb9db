// Minimal JSON reader for the scene / renderer files.
// Mirrors what the reference keeps of serde_json values in InputParamsValue
// (reference src/core/loader.rs:18-24, 412-438): Int vs Float are distinct
// (a literal without '.', 'e' or 'E' is an Int), Bool, String, Array, Object.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace spt_host {

struct JsonValue {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;  // insertion order kept

    const JsonValue* get(const std::string& key) const {
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_number() const { return kind == Int || kind == Float; }
    double as_double() const { return kind == Int ? (double)i : f; }
};

class JsonParser {
   public:
    explicit JsonParser(const std::string& text) : t_(text) {}
    JsonValue parse() {
        JsonValue v = value();
        ws();
        if (p_ != t_.size()) fail("trailing characters");
        return v;
    }

   private:
    const std::string& t_;
    size_t p_ = 0;

    [[noreturn]] void fail(const char* what) {
        size_t line = 1;
        for (size_t k = 0; k < p_ && k < t_.size(); ++k)
            if (t_[k] == '\n') ++line;
        throw std::runtime_error(std::string("json: ") + what + " at line " + std::to_string(line));
    }
    void ws() {
        while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) ++p_;
    }
    char peek() {
        ws();
        if (p_ >= t_.size()) fail("unexpected end");
        return t_[p_];
    }
    void expect(char c) {
        if (peek() != c) fail("unexpected character");
        ++p_;
    }
    JsonValue value() {
        char c = peek();
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') {
            JsonValue v;
            v.kind = JsonValue::String;
            v.s = string();
            return v;
        }
        if (c == 't' || c == 'f' || c == 'n') return literal();
        return number();
    }
    JsonValue literal() {
        JsonValue v;
        if (t_.compare(p_, 4, "true") == 0) {
            v.kind = JsonValue::Bool; v.b = true; p_ += 4;
        } else if (t_.compare(p_, 5, "false") == 0) {
            v.kind = JsonValue::Bool; v.b = false; p_ += 5;
        } else if (t_.compare(p_, 4, "null") == 0) {
            v.kind = JsonValue::Null; p_ += 4;
        } else {
            fail("bad literal");
        }
        return v;
    }
    JsonValue number() {
        size_t start = p_;
        bool is_float = false;
        if (p_ < t_.size() && (t_[p_] == '-' || t_[p_] == '+')) ++p_;
        while (p_ < t_.size()) {
            char c = t_[p_];
            if (c >= '0' && c <= '9') { ++p_; continue; }
            if (c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+') { is_float = true; ++p_; continue; }
            break;
        }
        if (p_ == start) fail("bad number");
        std::string tok = t_.substr(start, p_ - start);
        JsonValue v;
        if (is_float) {
            v.kind = JsonValue::Float;
            v.f = std::strtod(tok.c_str(), nullptr);
        } else {
            v.kind = JsonValue::Int;
            v.i = std::strtoll(tok.c_str(), nullptr, 10);
            v.f = (double)v.i;
        }
        return v;
    }
    std::string string() {
        expect('"');
        std::string out;
        while (true) {
            if (p_ >= t_.size()) fail("unterminated string");
            char c = t_[p_++];
            if (c == '"') break;
            if (c == '\\') {
                if (p_ >= t_.size()) fail("bad escape");
                char e = t_[p_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (p_ + 4 > t_.size()) fail("bad \\u escape");
                        unsigned cp = (unsigned)std::strtoul(t_.substr(p_, 4).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += e; break;
                }
            } else {
                out += c;
            }
        }
        return out;
    }
    JsonValue array() {
        expect('[');
        JsonValue v;
        v.kind = JsonValue::Array;
        if (peek() == ']') { ++p_; return v; }
        while (true) {
            v.arr.push_back(value());
            char c = peek();
            ++p_;
            if (c == ']') break;
            if (c != ',') fail("expected ',' or ']'");
        }
        return v;
    }
    JsonValue object() {
        expect('{');
        JsonValue v;
        v.kind = JsonValue::Object;
        if (peek() == '}') { ++p_; return v; }
        while (true) {
            peek();
            std::string k = string();
            expect(':');
            JsonValue val = value();
            bool replaced = false;
            for (auto& kv : v.obj)
                if (kv.first == k) { kv.second = val; replaced = true; }
            if (!replaced) v.obj.emplace_back(std::move(k), std::move(val));
            char c = peek();
            ++p_;
            if (c == '}') break;
            if (c != ',') fail("expected ',' or '}'");
        }
        return v;
    }
};

}  // namespace spt_host
