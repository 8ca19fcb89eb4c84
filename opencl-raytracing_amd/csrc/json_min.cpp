// json_min.cpp -- recursive-descent JSON reader (see json_min.hpp).
#include "json_min.hpp"

#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace jsonmin {

const Value* Value::find(const std::string& key) const {
    if (kind != Kind::Object) return nullptr;
    for (const auto& m : members)
        if (m.first == key) return &m.second;
    return nullptr;
}

size_t Value::size() const {
    switch (kind) {
        case Kind::Null: return 0;
        case Kind::Array: return items.size();
        case Kind::Object: return members.size();
        default: return 1;
    }
}

double Value::as_double(bool& ok) const {
    ok = true;
    switch (kind) {
        case Kind::Int: return static_cast<double>(i);
        case Kind::UInt: return static_cast<double>(u);
        case Kind::Float: return d;
        case Kind::Bool: return b ? 1.0 : 0.0;
        default: ok = false; return 0.0;
    }
}

std::vector<const Value*> Value::children() const {
    std::vector<const Value*> out;
    switch (kind) {
        case Kind::Null: break;
        case Kind::Array: for (const auto& v : items) out.push_back(&v); break;
        case Kind::Object: for (const auto& m : members) out.push_back(&m.second); break;
        default: out.push_back(this); break;
    }
    return out;
}

namespace {

struct Reader {
    const char* p;
    const char* end;
    const char* begin;
    std::string err;
    int depth = 0;

    bool fail(const char* what) {
        if (err.empty()) err = std::string(what) + " at byte " + std::to_string(p - begin);
        return false;
    }
    void skip_ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool literal(const char* word) {
        size_t n = std::strlen(word);
        if (static_cast<size_t>(end - p) < n || std::memcmp(p, word, n) != 0) return fail("bad literal");
        p += n;
        return true;
    }
    static void put_utf8(std::string& s, uint32_t cp) {
        if (cp < 0x80) s.push_back(static_cast<char>(cp));
        else if (cp < 0x800) {
            s.push_back(static_cast<char>(0xC0 | (cp >> 6)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            s.push_back(static_cast<char>(0xE0 | (cp >> 12)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else {
            s.push_back(static_cast<char>(0xF0 | (cp >> 18)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t& out) {
        if (end - p < 4) return fail("short \\u escape");
        out = 0;
        for (int k = 0; k < 4; ++k) {
            char c = *p++;
            out <<= 4;
            if (c >= '0' && c <= '9') out |= static_cast<uint32_t>(c - '0');
            else if (c >= 'a' && c <= 'f') out |= static_cast<uint32_t>(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') out |= static_cast<uint32_t>(c - 'A' + 10);
            else return fail("bad hex digit");
        }
        return true;
    }
    bool string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        out.clear();
        while (true) {
            if (p >= end) return fail("unterminated string");
            unsigned char c = static_cast<unsigned char>(*p++);
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') { out.push_back(static_cast<char>(c)); continue; }
            if (p >= end) return fail("unterminated escape");
            char e = *p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp = 0;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        if (end - p < 2 || p[0] != '\\' || p[1] != 'u') return fail("lone surrogate");
                        p += 2;
                        uint32_t lo = 0;
                        if (!hex4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) return fail("bad low surrogate");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                        return fail("lone surrogate");
                    }
                    put_utf8(out, cp);
                    break;
                }
                default: return fail("bad escape");
            }
        }
    }
    bool number(Value& v) {
        const char* s = p;
        bool neg = false, is_float = false;
        if (p < end && *p == '-') { neg = true; ++p; }
        if (p >= end) return fail("bad number");
        if (*p == '0') ++p;
        else if (*p >= '1' && *p <= '9') { while (p < end && *p >= '0' && *p <= '9') ++p; }
        else return fail("bad number");
        if (p < end && *p == '.') {
            is_float = true;
            ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad fraction");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            is_float = true;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad exponent");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        // The token is already validated; from_chars converts it in place, correctly rounded.
        if (!is_float) {
            if (neg) {
                long long x = 0;
                auto r = std::from_chars(s, p, x);
                if (r.ec == std::errc() && r.ptr == p) { v.kind = Kind::Int; v.i = x; return true; }
            } else {
                unsigned long long x = 0;
                auto r = std::from_chars(s, p, x);
                if (r.ec == std::errc() && r.ptr == p) { v.kind = Kind::UInt; v.u = x; return true; }
            }
            // Out of 64-bit range: the token becomes a float.
        }
        v.kind = Kind::Float;
        double x = 0.0;
        auto r = std::from_chars(s, p, x);
        if (r.ec == std::errc::result_out_of_range) {
            // strtod semantics: +-HUGE_VAL on overflow, a zero or denormal on underflow
            std::string tok(s, p);
            x = std::strtod(tok.c_str(), nullptr);
        }
        // The DOM library the reference uses refuses a literal that overflows double
        // ("number overflow parsing"), so it is a parse error here too.
        if (!std::isfinite(x)) { p = s; return fail("number overflow"); }
        v.d = x;
        return true;
    }
    bool value(Value& v) {
        if (++depth > 512) return fail("nesting too deep");
        skip_ws();
        if (p >= end) return fail("unexpected end of input");
        bool ok = false;
        switch (*p) {
            case '{': ok = object(v); break;
            case '[': ok = array(v); break;
            case '"': v.kind = Kind::String; ok = string(v.s); break;
            case 't': ok = literal("true"); v.kind = Kind::Bool; v.b = true; break;
            case 'f': ok = literal("false"); v.kind = Kind::Bool; v.b = false; break;
            case 'n': ok = literal("null"); v.kind = Kind::Null; break;
            default: ok = number(v); break;
        }
        --depth;
        return ok;
    }
    bool array(Value& v) {
        v.kind = Kind::Array;
        ++p;
        skip_ws();
        if (p < end && *p == ']') { ++p; return true; }
        v.items.reserve(4);
        while (true) {
            v.items.emplace_back();
            if (!value(v.items.back())) return false;
            skip_ws();
            if (p >= end) return fail("unterminated array");
            if (*p == ',') { ++p; continue; }
            if (*p == ']') { ++p; return true; }
            return fail("expected ',' or ']'");
        }
    }
    bool object(Value& v) {
        v.kind = Kind::Object;
        ++p;
        skip_ws();
        if (p < end && *p == '}') { ++p; return true; }
        v.members.reserve(8);
        while (true) {
            skip_ws();
            std::string key;
            if (!string(key)) return false;
            skip_ws();
            if (p >= end || *p != ':') return fail("expected ':'");
            ++p;
            Value child;
            if (!value(child)) return false;
            bool replaced = false;
            for (auto& m : v.members)
                if (m.first == key) { m.second = std::move(child); replaced = true; break; }
            if (!replaced) v.members.emplace_back(std::move(key), std::move(child));
            skip_ws();
            if (p >= end) return fail("unterminated object");
            if (*p == ',') { ++p; continue; }
            if (*p == '}') { ++p; return true; }
            return fail("expected ',' or '}'");
        }
    }
};

}  // namespace

bool parse(const char* text, size_t len, Value& out, std::string& err) {
    Reader r{text, text + len, text, std::string(), 0};
    out = Value();
    if (!r.value(out)) { err = r.err; return false; }
    return true;
}

}  // namespace jsonmin
