// json.h — a small recursive-descent JSON reader (objects, arrays, strings, numbers, true/false/null), enough
// for HF config.json / generation_config.json, the safetensors header and model.safetensors.index.json.
// The reference uses rapidjson for the same files (src/huggingface/JsonHelper.h, src/util/SafeTensors.cpp:150).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace tgxh {

struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  bool is_int = false;
  int64_t i = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;   // insertion order kept (safetensors header order)

  const Json* get(const char* key) const {
    if (kind != Obj) return nullptr;
    for (const auto& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool has(const char* key) const { return get(key) != nullptr; }
  // typed getters with defaults == getJsonValue<T> (JsonHelper.h:13-35)
  int64_t get_int(const char* key, int64_t def) const { const Json* v = get(key); return (v && v->kind == Num && v->is_int) ? v->i : def; }
  float get_float(const char* key, float def) const { const Json* v = get(key); return (v && v->kind == Num) ? (float)v->num : def; }
  bool get_bool(const char* key, bool def) const { const Json* v = get(key); return (v && v->kind == Bool) ? v->b : def; }
  std::string get_str(const char* key, const std::string& def) const { const Json* v = get(key); return (v && v->kind == Str) ? v->str : def; }
};

class JsonParser {
 public:
  // returns false on a syntax error (the reference logs "JSON parse error" and fails the load)
  static bool parse(const char* s, size_t n, Json& out) {
    JsonParser p(s, n);
    p.ws();
    if (!p.value(out)) return false;
    p.ws();
    return p.pos_ == p.n_;
  }

 private:
  JsonParser(const char* s, size_t n) : s_(s), n_(n) {}
  const char* s_; size_t n_; size_t pos_ = 0;
  int depth_ = 0;                                  // nesting of the value being parsed; untrusted headers must not overflow the stack
  static constexpr int kMaxDepth = 64;
  void ws() { while (pos_ < n_ && (s_[pos_] == ' ' || s_[pos_] == '\n' || s_[pos_] == '\t' || s_[pos_] == '\r')) pos_++; }
  bool lit(const char* w) { size_t l = strlen(w); if (pos_ + l <= n_ && !memcmp(s_ + pos_, w, l)) { pos_ += l; return true; } return false; }
  bool value(Json& o) {
    if (pos_ >= n_) return false;
    char c = s_[pos_];
    if (c == '{' || c == '[') {
      if (depth_ >= kMaxDepth) return false;
      depth_++;
      const bool ok = c == '{' ? object(o) : array(o);
      depth_--;
      return ok;
    }
    if (c == '"') { o.kind = Json::Str; return string(o.str); }
    if (lit("true")) { o.kind = Json::Bool; o.b = true; return true; }
    if (lit("false")) { o.kind = Json::Bool; o.b = false; return true; }
    if (lit("null")) { o.kind = Json::Null; return true; }
    return number(o);
  }
  bool number(Json& o) {
    size_t st = pos_;
    bool isint = true;
    if (pos_ < n_ && (s_[pos_] == '-' || s_[pos_] == '+')) pos_++;
    while (pos_ < n_) {
      char c = s_[pos_];
      if (c >= '0' && c <= '9') pos_++;
      else if (c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+') { isint = false; pos_++; }
      else break;
    }
    if (pos_ == st) return false;
    std::string t(s_ + st, pos_ - st);
    // JSON as written by python may contain NaN/Infinity; not needed for the files read here
    o.kind = Json::Num; o.is_int = isint; o.num = strtod(t.c_str(), nullptr);
    if (isint) o.i = strtoll(t.c_str(), nullptr, 10);
    return true;
  }
  bool string(std::string& out) {
    if (s_[pos_] != '"') return false;
    pos_++;
    out.clear();
    while (pos_ < n_ && s_[pos_] != '"') {
      char c = s_[pos_++];
      if (c == '\\' && pos_ < n_) {
        char e = s_[pos_++];
        switch (e) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break;
          case 'u': {
            if (pos_ + 4 > n_) return false;
            unsigned cp = (unsigned)strtoul(std::string(s_ + pos_, 4).c_str(), nullptr, 16);
            pos_ += 4;
            if (cp >= 0xD800 && cp < 0xDC00 && pos_ + 6 <= n_ && s_[pos_] == '\\' && s_[pos_ + 1] == 'u') {   // surrogate pair
              const unsigned lo = (unsigned)strtoul(std::string(s_ + pos_ + 2, 4).c_str(), nullptr, 16);
              if (lo >= 0xDC00 && lo < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); pos_ += 6; }
            }
            if (cp >= 0x10000) {
              out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
            } else
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e;
        }
      } else out += c;
    }
    if (pos_ >= n_) return false;
    pos_++;
    return true;
  }
  bool array(Json& o) {
    o.kind = Json::Arr; pos_++; ws();
    if (pos_ < n_ && s_[pos_] == ']') { pos_++; return true; }
    while (true) {
      o.arr.emplace_back();
      ws(); if (!value(o.arr.back())) return false; ws();
      if (pos_ < n_ && s_[pos_] == ',') { pos_++; continue; }
      if (pos_ < n_ && s_[pos_] == ']') { pos_++; return true; }
      return false;
    }
  }
  bool object(Json& o) {
    o.kind = Json::Obj; pos_++; ws();
    if (pos_ < n_ && s_[pos_] == '}') { pos_++; return true; }
    while (true) {
      ws(); std::string k; if (pos_ >= n_ || !string(k)) return false;
      ws(); if (pos_ >= n_ || s_[pos_] != ':') return false; pos_++; ws();
      o.obj.emplace_back(k, Json());
      if (!value(o.obj.back().second)) return false;
      ws();
      if (pos_ < n_ && s_[pos_] == ',') { pos_++; continue; }
      if (pos_ < n_ && s_[pos_] == '}') { pos_++; return true; }
      return false;
    }
  }
};

}  // namespace tgxh
