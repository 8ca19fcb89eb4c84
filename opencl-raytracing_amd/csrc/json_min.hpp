// json_min.hpp -- a small strict JSON reader for scene files.
//
// The reference reads scenes through nlohmann/json (rayson.hpp:24, a brew
// dependency that is not vendored).  Only the behaviour the loader depends on
// is reproduced: strict RFC 8259 syntax, the integer / unsigned / float
// distinction of number tokens (rayson.hpp:404-430 tests is_number_float and
// is_number_integer), last-one-wins for repeated object keys, and "parse one
// value, ignore what follows" (stream extraction, rayson.hpp:629).
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace jsonmin {

enum class Kind : uint8_t { Null, Bool, Int, UInt, Float, String, Array, Object };

struct Value {
    Kind kind = Kind::Null;
    bool b = false;
    int64_t i = 0;      // Kind::Int  (a negative integer literal)
    uint64_t u = 0;     // Kind::UInt (a non-negative integer literal)
    double d = 0.0;     // Kind::Float
    std::string s;      // Kind::String
    std::vector<Value> items;                          // Kind::Array
    std::vector<std::pair<std::string, Value>> members; // Kind::Object, unique keys

    bool is_object() const { return kind == Kind::Object; }
    bool is_array() const { return kind == Kind::Array; }
    bool is_string() const { return kind == Kind::String; }
    bool is_boolean() const { return kind == Kind::Bool; }
    bool is_number_float() const { return kind == Kind::Float; }
    bool is_number_integer() const { return kind == Kind::Int || kind == Kind::UInt; }
    bool is_number() const { return is_number_float() || is_number_integer(); }

    // Member lookup; nullptr when this is not an object or the key is absent.
    const Value* find(const std::string& key) const;
    // Element count the way the DOM library reports it: 0 for null, 1 for
    // scalars, the length for arrays and objects.
    size_t size() const;
    // Numeric value of a number or bool; ok=false for any other kind.
    double as_double(bool& ok) const;
    // Children to walk in a range-for: array elements, object values, nothing
    // for null, the value itself for a scalar.
    std::vector<const Value*> children() const;
};

// Parses the first JSON value in [text, text+len).  Returns false on a syntax
// error (err gets a short description with the byte offset).
bool parse(const char* text, size_t len, Value& out, std::string& err);

}  // namespace jsonmin
