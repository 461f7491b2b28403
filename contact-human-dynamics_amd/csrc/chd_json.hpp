// chd_json.hpp -- a small JSON reader (host) for the two input formats in front of the kinematic optimisation: OpenPose's per-frame result files
// (openpose_utils.py:48-76 reads people[0].pose_keypoints_2d) and monocular total capture's tracked_results.json (totalcap_utils.py:33-79).
//
// Written from RFC 8259: objects, arrays, strings (escapes are VALIDATED -- the nine escape characters, four hexadecimal digits after \u, well-formed UTF-8 -- but not decoded:
// no value read here is a string), numbers through std::from_chars / strtod (correctly rounded, like Python's float), true / false / null, and -- as Python's json module does
// beyond the RFC -- the literals NaN, Infinity and -Infinity.  Strict otherwise: trailing commas, a second top-level value, duplicate keys are errors naming the byte offset; the
// caller reports it with the file name and fails that file only.  The Python mirrors (contact_net.load_keypoint_dir, totalcap_io.load_totalcap_results) use the json
// module; tests/test_ingest_native.py holds the two together, value for value.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <charconv>

#include <string>
#include <vector>

namespace chd_json {

struct Key { const char* s; int n; };            // a view into the parsed text (no escapes: see Parser::string), which outlives the tree
struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  double num = 0.0;
  bool flag = false;
  bool all_numbers = false;                       // an array whose elements are all numbers keeps them in `nums` (one allocation instead of a Value per element:
  std::vector<double> nums;                       //   the coefficient arrays and OpenPose's keypoint lists are most of what these files hold)
  std::vector<Value> items;                       // other array elements / object values
  std::vector<Key> keys;                          // object keys, parallel to items
  const Value* get(const char* key) const {       // the member of that name (duplicates are refused at parse time: Python's dict would keep the last)
    if (kind != Object) return nullptr;
    const int n = (int)strlen(key);
    for (size_t i = 0; i < keys.size(); ++i) if (keys[i].n == n && !memcmp(keys[i].s, key, n)) return &items[i];
    return nullptr;
  }
  size_t size() const { return all_numbers ? nums.size() : items.size(); }
};

struct Parser {
  const char* p; const char* e; const char* b;
  std::string err;
  bool fail(const char* what) { if (err.empty()) err = std::string(what) + " at byte " + std::to_string((long long)(p - b)); return false; }
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  bool string(Key* out) {
    if (p >= e || *p != '"') return fail("expected a string");
    ++p;
    const char* s = p;
    bool escaped = false;
    while (p < e && *p != '"') {
      const unsigned char ch = (unsigned char)*p;
      if (ch < 0x20) return fail("control character in a string");
      if (ch == '\\') {
        escaped = true; ++p;
        if (p >= e) break;
        if (*p == 'u') {
          if (e - p < 5) return fail("truncated \\u escape");
          for (int k = 1; k <= 4; ++k) { const char h = p[k]; if (!((h >= '0' && h <= '9') || (h >= 'a' && h <= 'f') || (h >= 'A' && h <= 'F'))) return fail("\\u escape needs four hexadecimal digits"); }
          p += 4;
        } else if (!strchr("\"\\/bfnrt", *p)) return fail("invalid escape in a string");
        ++p;
        continue;
      }
      if (ch >= 0x80) {          // UTF-8, as Python's text-mode reader demands it (well-formed sequences only: no stray continuation bytes, overlong forms, surrogates, or code points above U+10FFFF)
        int n = ch >= 0xF0 ? 3 : ch >= 0xE0 ? 2 : 1;
        if (ch < 0xC2 || ch > 0xF4 || e - p <= n) return fail("invalid UTF-8 in a string");
        for (int k = 1; k <= n; ++k) if (((unsigned char)p[k] & 0xC0) != 0x80) return fail("invalid UTF-8 in a string");
        const unsigned char c1 = (unsigned char)p[1];
        if ((ch == 0xE0 && c1 < 0xA0) || (ch == 0xED && c1 > 0x9F) || (ch == 0xF0 && c1 < 0x90) || (ch == 0xF4 && c1 > 0x8F)) return fail("invalid UTF-8 in a string");
        p += n;
      }
      ++p;
    }
    if (p >= e) return fail("unterminated string");
    if (out) { if (escaped) return fail("escape sequence in an object key"); out->s = s; out->n = (int)(p - s); }
    ++p;
    return true;
  }
  bool number(double* out) {
    // the grammar first (strtod alone would take "0x10", "inf", ".5"), then the conversion
    const char* s = p;
    // the three literals Python's json module accepts beside the RFC's grammar (trackers occasionally emit them): the mirrors read them, so does this reader
    if (e - p >= 3 && !memcmp(p, "NaN", 3)) { p += 3; *out = __builtin_nan(""); return true; }
    if (e - p >= 8 && !memcmp(p, "Infinity", 8)) { p += 8; *out = __builtin_inf(); return true; }
    if (e - p >= 9 && !memcmp(p, "-Infinity", 9)) { p += 9; *out = -__builtin_inf(); return true; }
    if (p < e && *p == '-') ++p;
    if (p >= e) return fail("truncated number");
    if (*p == '0') ++p;
    else if (*p >= '1' && *p <= '9') { while (p < e && *p >= '0' && *p <= '9') ++p; }
    else return fail("expected a value");
    bool integer = true;                           // (Python reads such a token as an int: "-0" is then 0, i.e. +0.0 as a float)
    if (p < e && *p == '.') { integer = false; ++p; if (p >= e || *p < '0' || *p > '9') return fail("digits expected after the decimal point"); while (p < e && *p >= '0' && *p <= '9') ++p; }
    if (p < e && (*p == 'e' || *p == 'E')) {
      integer = false;
      ++p;
      if (p < e && (*p == '+' || *p == '-')) ++p;
      if (p >= e || *p < '0' || *p > '9') return fail("digits expected in the exponent");
      while (p < e && *p >= '0' && *p <= '9') ++p;
    }
    // std::from_chars: correctly rounded like strtod (and Python's float), several times faster; it reports overflow AND underflow as out of range without a value: strtod then
    // supplies the infinity / zero / subnormal the other two would give
    const std::from_chars_result fr = std::from_chars(s, p, *out);
    if (fr.ec != std::errc() || fr.ptr != p) {
      const std::string t(s, (size_t)(p - s));
      *out = strtod(t.c_str(), nullptr);
    }
    if (integer && *out == 0.0) *out = 0.0;
    return true;
  }
  bool value(Value& v, int depth) {
    if (depth > 64) return fail("nesting deeper than 64");
    ws();
    if (p >= e) return fail("unexpected end of the text");
    const char c = *p;
    if (c == '{') {
      v.kind = Value::Object; ++p; ws();
      if (p < e && *p == '}') { ++p; return true; }
      for (;;) {
        ws();
        Key k{nullptr, 0};
        if (!string(&k)) return false;
        for (const Key& q : v.keys) if (q.n == k.n && !memcmp(q.s, k.s, k.n)) return fail("duplicate key in an object");
        ws();
        if (p >= e || *p != ':') return fail("expected ':'");
        ++p;
        v.keys.push_back(k); v.items.emplace_back();
        if (!value(v.items.back(), depth + 1)) return false;
        ws();
        if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.kind = Value::Array; ++p; ws();
      if (p < e && *p == ']') { ++p; return true; }
      v.all_numbers = true;
      for (;;) {
        ws();
        if (v.all_numbers && p < e && (*p == '-' || (*p >= '0' && *p <= '9') || *p == 'N' || *p == 'I')) {
          double x;
          if (!number(&x)) return false;
          v.nums.push_back(x);
        } else {
          if (v.all_numbers) {                    // (a mixed array after all: the numbers so far become elements like the others)
            v.all_numbers = false;
            for (double x : v.nums) { v.items.emplace_back(); v.items.back().kind = Value::Number; v.items.back().num = x; }
            v.nums.clear();
          }
          v.items.emplace_back();
          if (!value(v.items.back(), depth + 1)) return false;
        }
        ws();
        if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') { v.kind = Value::String; return string(nullptr); }
    if (e - p >= 4 && !memcmp(p, "true", 4)) { v.kind = Value::Bool; v.flag = true; p += 4; return true; }
    if (e - p >= 5 && !memcmp(p, "false", 5)) { v.kind = Value::Bool; p += 5; return true; }
    if (e - p >= 4 && !memcmp(p, "null", 4)) { v.kind = Value::Null; p += 4; return true; }
    v.kind = Value::Number;
    return number(&v.num);
  }
};

// "" on success, else the reason
inline std::string parse(const std::string& text, Value& root) {
  Parser ps{text.data(), text.data() + text.size(), text.data(), {}};
  if (!ps.value(root, 0)) return ps.err;
  ps.ws();
  if (ps.p != ps.e) { ps.fail("text after the top-level value"); return ps.err; }
  return "";
}

// an array of numbers appended to `out`; false if it is anything else
inline bool numbers(const Value* a, std::vector<double>& out, size_t expect = (size_t)-1) {
  if (!a || a->kind != Value::Array || !(a->all_numbers || a->items.empty()) || (expect != (size_t)-1 && a->nums.size() != expect)) return false;
  out.insert(out.end(), a->nums.begin(), a->nums.end());
  return true;
}
// {"x":..,"y":..,"z":..} appended to `out`
inline bool xyz(const Value* o, std::vector<double>& out) {
  if (!o) return false;
  const Value *x = o->get("x"), *y = o->get("y"), *z = o->get("z");
  if (!x || !y || !z || x->kind != Value::Number || y->kind != Value::Number || z->kind != Value::Number) return false;
  out.push_back(x->num); out.push_back(y->num); out.push_back(z->num);
  return true;
}

}  // namespace chd_json
