// tokenizer.cpp — see tokenizer.h.  Behaviour follows tinygpt::tokenizer (src/tokenizer/*.cpp, cited per function);
// the structure is this repo's own: a tokenizer.json is compiled into three flat stage lists (normalizer,
// pre-tokenizer, decoder) over plain strings, BPE runs on token ids with a rank heap (O(n log n) for any piece
// length), added tokens are matched leftmost-longest.
#include "tokenizer.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <queue>
#include <sstream>
#include <thread>

#include "json.h"
#include "unicode_tables.h"

namespace tgxh {

namespace {

const char* kGpt2Pattern = R"('s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+)";   // openai/gpt-2 encoder.py

// GPT-2 bytes_to_unicode: printable Latin-1 bytes map to themselves, the other 68 bytes to U+0100..U+0143
struct ByteMap {
  uint32_t cp_of_byte[256];
  int byte_of_cp[0x144];
  ByteMap() {
    for (int& b : byte_of_cp) b = -1;
    int n = 0;
    for (int b = 0; b < 256; b++) {
      const bool keep = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
      cp_of_byte[b] = keep ? (uint32_t)b : (uint32_t)(256 + n++);
      byte_of_cp[cp_of_byte[b]] = b;
    }
  }
};
const ByteMap& byte_map() { static const ByteMap m; return m; }

std::string bytes_to_unicode(const std::string& raw) {
  std::string out;
  out.reserve(raw.size() * 2);
  for (unsigned char c : raw) utf8_append(out, byte_map().cp_of_byte[c]);
  return out;
}
// inverse: code points of the byte-level alphabet become bytes, anything else is kept (== ByteLevel::utf8ToBytes)
std::string unicode_to_bytes(const std::string& s) {
  std::string out;
  out.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    uint32_t cp;
    const size_t n = utf8_decode(s.data() + i, s.size() - i, cp);
    if (cp < 0x144 && byte_map().byte_of_cp[cp] >= 0) out += (char)byte_map().byte_of_cp[cp];
    else out.append(s, i, n);
    i += n;
  }
  return out;
}

// position of the first byte of an incomplete UTF-8 sequence at the end of s, or npos (== ByteLevel::findIncompletePos)
size_t incomplete_utf8_tail(const std::string& s) {
  const size_t n = s.size();
  for (size_t back = 1; back <= 4 && back <= n; back++) {
    const unsigned char c = (unsigned char)s[n - back];
    if ((c & 0xC0) == 0x80) continue;                       // continuation byte: keep looking for the lead
    const size_t need = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : c >= 0xC0 ? 2 : 1;
    return need > back ? n - back : std::string::npos;
  }
  return std::string::npos;
}

bool valid_utf8(const std::string& s) {
  for (size_t i = 0; i < s.size();) {
    uint32_t cp;
    const size_t n = utf8_decode(s.data() + i, s.size() - i, cp);
    if (cp == 0xFFFD && !(n == 3 && (unsigned char)s[i] == 0xEF)) return false;
    i += n;
  }
  return true;
}

// ---- Unicode normalisation: NF(K)D = full (compatibility) decomposition + canonical ordering; NF(K)C = that + composition
int ccc_of(uint32_t cp) {
  int lo = 0, hi = unidata::kCccCount - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) / 2;
    if (cp < unidata::kCcc[mid].cp) hi = mid - 1;
    else if (cp > unidata::kCcc[mid].cp) lo = mid + 1;
    else return unidata::kCcc[mid].cls;
  }
  return 0;
}
void decompose(uint32_t cp, std::vector<uint32_t>& out, bool compat) {
  if (compat) {                          // full NFKD of the code points whose compatibility mapping differs from the canonical one
    int lo = 0, hi = unidata::kCompatCount - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) / 2;
      if (cp < unidata::kCompat[mid].cp) hi = mid - 1;
      else if (cp > unidata::kCompat[mid].cp) lo = mid + 1;
      else { for (int k = 0; k < unidata::kCompat[mid].len; k++) out.push_back(unidata::kCompatData[unidata::kCompat[mid].off + k]); return; }
    }
  }
  if (cp >= 0xAC00 && cp <= 0xD7A3) {   // Hangul syllable -> L V (T)
    const uint32_t s = cp - 0xAC00, t = s % 28;
    out.push_back(0x1100 + s / 588); out.push_back(0x1161 + (s % 588) / 28);
    if (t) out.push_back(0x11A7 + t);
    return;
  }
  int lo = 0, hi = unidata::kDecompCount - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) / 2;
    if (cp < unidata::kDecomp[mid].cp) hi = mid - 1;
    else if (cp > unidata::kDecomp[mid].cp) lo = mid + 1;
    else { for (int k = 0; k < unidata::kDecomp[mid].len; k++) out.push_back(unidata::kDecompData[unidata::kDecomp[mid].off + k]); return; }
  }
  out.push_back(cp);
}
uint32_t compose_pair(uint32_t a, uint32_t b) {
  if (a >= 0x1100 && a < 0x1113 && b >= 0x1161 && b < 0x1176) return 0xAC00 + ((a - 0x1100) * 21 + (b - 0x1161)) * 28;
  if (a >= 0xAC00 && a <= 0xD7A3 && (a - 0xAC00) % 28 == 0 && b > 0x11A7 && b < 0x11C3) return a + (b - 0x11A7);
  int lo = 0, hi = unidata::kComposeCount - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) / 2;
    const unidata::Compose& c = unidata::kCompose[mid];
    if (a < c.a || (a == c.a && b < c.b)) hi = mid - 1;
    else if (a > c.a || (a == c.a && b > c.b)) lo = mid + 1;
    else return c.c;
  }
  return 0;
}
}  // namespace
std::string normalize_unicode(const std::string& text, bool compose, bool compat) {
  bool ascii = true, plain = true;
  for (unsigned char c : text) { if (c >= 0x80) ascii = false; if (c >= 0xCC) { plain = false; break; } }
  if (ascii) return text;                                                  // ASCII is stable under all four forms
  if (plain && compose && !compat) return text;                            // below U+0300 precomposed text is already NFC
  std::vector<uint32_t> cps;
  for (size_t i = 0; i < text.size();) { uint32_t cp; i += utf8_decode(text.data() + i, text.size() - i, cp); decompose(cp, cps, compat); }
  for (size_t i = 1; i < cps.size(); i++) {          // canonical ordering: stable sort of each run of non-starters by class
    const int c = ccc_of(cps[i]);
    if (!c) continue;
    size_t j = i;
    while (j > 0 && ccc_of(cps[j - 1]) > c) { std::swap(cps[j], cps[j - 1]); j--; }
  }
  if (compose && !cps.empty()) {
    std::vector<uint32_t> out;
    out.push_back(cps[0]);
    size_t starter = 0;
    int last_ccc = ccc_of(cps[0]) ? 256 : 0;         // a leading non-starter can never be composed onto
    for (size_t i = 1; i < cps.size(); i++) {
      const uint32_t c = cps[i];
      const int cc = ccc_of(c);
      const uint32_t comp = (last_ccc < cc || last_ccc == 0) ? compose_pair(out[starter], c) : 0;   // not blocked from the last starter
      if (comp) { out[starter] = comp; continue; }
      if (cc == 0) starter = out.size();
      last_ccc = cc;
      out.push_back(c);
    }
    cps.swap(out);
  }
  std::string res;
  res.reserve(text.size());
  for (uint32_t cp : cps) utf8_append(res, cp);
  return res;
}
namespace {

std::string read_file(const std::string& path, bool& ok) {
  std::ifstream in(path, std::ios::binary);
  ok = (bool)in;
  if (!ok) return {};
  std::ostringstream ss;
  ss << in.rdbuf();
  return ss.str();
}

std::string quote_meta(const std::string& s) {
  std::string out;
  for (char c : s) {
    if (strchr("\\^$.|?*+()[]{}-", c)) out += '\\';
    out += c;
  }
  return out;
}

enum class Behavior { Removed, Isolated, MergedWithPrevious, MergedWithNext, Contiguous };

// Split::split (Split.cpp:44-160): the five delimiter behaviours over the match list of one piece
void split_by_matches(const std::string& s, const std::vector<Range>& matches, Behavior how, std::vector<std::string>& out) {
  std::vector<Range> r;
  size_t pos = 0;
  Range last(0, 0);
  for (const Range& m : matches) {
    switch (how) {
      case Behavior::Removed: if (m.first > pos) r.emplace_back(pos, m.first); break;
      case Behavior::Isolated: if (m.first > pos) r.emplace_back(pos, m.first); r.push_back(m); break;
      case Behavior::MergedWithPrevious: r.emplace_back(m.first > pos ? pos : m.first, m.second); break;
      case Behavior::MergedWithNext:     // a match opens a piece that runs up to the next match
        if (m.first > pos) { if (pos == last.second && !r.empty()) r.back() = Range(last.first, m.first); else r.emplace_back(pos, m.first); }
        r.push_back(m); last = m; break;
      case Behavior::Contiguous:         // adjacent matches fuse into one piece
        if (m.first > pos) r.emplace_back(pos, m.first);
        if (m.first == pos && !r.empty()) r.back().second = m.second; else r.push_back(m);
        break;
    }
    pos = m.second;
  }
  if (s.size() > pos) {
    if (how == Behavior::MergedWithNext && pos == last.second && !r.empty()) r.back() = Range(last.first, s.size());
    else r.emplace_back(pos, s.size());
  }
  for (const Range& x : r) out.emplace_back(s, x.first, x.second - x.first);
}

}  // namespace

struct Tokenizer::Step {
  enum Type { NFC, NFD, NFKC, NFKD, ByteLevel, Split, Metaspace, Replace, ByteFallback, Fuse, Strip } type;
  bool flag_a = false, flag_b = false;           // ByteLevel: add_prefix_space, use_regex; Metaspace: split
  std::shared_ptr<Regex> regex;                  // Split / ByteLevel(use_regex)
  std::string str_a, str_b;                      // Split: literal pattern; Metaspace: replacement, scheme; Replace: from, to; Strip: content
  Behavior behavior = Behavior::Removed;
  bool invert = false;
  int start = 0, stop = 0;                       // Strip
};

Tokenizer::Tokenizer() { for (int32_t& b : byteTokenId_) b = -1; }
Tokenizer::~Tokenizer() = default;

// one component object (or a Sequence of them) -> flat stage list (TokenizerConfig.cpp:245-285, createComponent :468-500)
bool Tokenizer::parseSteps(const void* jv, std::vector<Step>& out, const char* what) {
  const Json& j = *static_cast<const Json*>(jv);
  if (j.kind == Json::Null) return true;
  if (j.kind != Json::Obj) return fail(std::string(what) + ": component is not an object");
  const std::string type = j.get_str("type", "");
  if (type == "Sequence") {
    for (const auto& kv : j.obj)
      if (kv.second.kind == Json::Arr)
        for (const Json& sub : kv.second.arr)
          if (!parseSteps(&sub, out, what)) return false;
    return true;
  }
  Step s;
  if (type == "NFC") s.type = Step::NFC;
  else if (type == "NFD") s.type = Step::NFD;
  else if (type == "NFKC") s.type = Step::NFKC;
  else if (type == "NFKD") s.type = Step::NFKD;
  else if (type == "ByteLevel") {
    s.type = Step::ByteLevel;
    s.flag_a = j.get_bool("add_prefix_space", false);
    // absent == false, as the reference reads it (TokenizerConfig.cpp:70).  HF `tokenizers` defaults an absent key to true (gpt2's
    // tokenizer.json has none): identical ids on ordinary text, different ones on whitespace runs (" \u00a0\u00a0", "a  b"), where the
    // unsplit text lets BPE merge across what GPT-2's pattern would have separated.  TGX_TOKENIZER_HF_DEFAULTS=1 selects HF's default.
    const char* hf_defaults = getenv("TGX_TOKENIZER_HF_DEFAULTS");
    s.flag_b = j.get_bool("use_regex", hf_defaults && hf_defaults[0] == '1');
    if (s.flag_b) s.regex = std::make_shared<Regex>(kGpt2Pattern);
    byteLevelDecode_ = true;
  } else if (type == "Split") {
    s.type = Step::Split;
    const Json* pat = j.get("pattern");
    std::string rx;
    if (pat && pat->kind == Json::Obj && !pat->obj.empty() && pat->obj[0].second.kind == Json::Str)
      rx = pat->obj[0].first == "String" ? quote_meta(pat->obj[0].second.str) : pat->obj[0].second.str;
    else if (pat && pat->kind == Json::Str) rx = pat->str;
    s.regex = std::make_shared<Regex>(rx);
    if (!s.regex->valid()) return fail(std::string(what) + ": Split pattern not supported (" + s.regex->error() + "): " + rx);
    const std::string b = j.get_str("behavior", "Removed");
    if (b == "Removed") s.behavior = Behavior::Removed;
    else if (b == "Isolated") s.behavior = Behavior::Isolated;
    else if (b == "MergedWithPrevious") s.behavior = Behavior::MergedWithPrevious;
    else if (b == "MergedWithNext") s.behavior = Behavior::MergedWithNext;
    else if (b == "Contiguous") s.behavior = Behavior::Contiguous;
    else return fail(std::string(what) + ": unknown Split behavior " + b);
    s.invert = j.get_bool("invert", false);
    if (s.invert) return fail(std::string(what) + ": Split invert mode is not supported (nor by the reference, Split.cpp:17-20)");
  } else if (type == "Metaspace") {
    s.type = Step::Metaspace;
    s.str_a = j.get_str("replacement", "\xE2\x96\x81");
    s.str_b = j.get_str("prepend_scheme", "always");
    s.flag_a = j.get_bool("split", true);
  } else if (type == "Replace") {
    s.type = Step::Replace;
    const Json* pat = j.get("pattern");
    if (!pat || pat->kind != Json::Obj || !pat->has("String")) return fail(std::string(what) + ": Replace supports String patterns only");
    s.str_a = pat->get_str("String", "");
    s.str_b = j.get_str("content", "");
  } else if (type == "ByteFallback") s.type = Step::ByteFallback;
  else if (type == "Fuse") s.type = Step::Fuse;
  else if (type == "Strip") {
    s.type = Step::Strip;
    s.str_a = j.get_str("content", "");
    s.start = (int)j.get_int("start", 0); s.stop = (int)j.get_int("stop", 0);
    if (s.str_a.size() != 1) return fail(std::string(what) + ": Strip content must be one byte");
  } else if (type == "TemplateProcessing") {
    // "single": specials before / after sequence A (TemplateProcessing.cpp:29-84; the pair template is unused there too)
    const Json* single = j.get("single");
    const Json* specials = j.get("special_tokens");
    bool after = false;
    if (single && single->kind == Json::Arr)
      for (const Json& e : single->arr) {
        if (e.has("Sequence")) { after = true; continue; }
        const Json* st = e.get("SpecialToken");
        if (!st) continue;
        const std::string id = st->get_str("id", "");
        const Json* def = specials ? specials->get(id.c_str()) : nullptr;
        const Json* ids = def ? def->get("ids") : nullptr;
        if (!ids || ids->kind != Json::Arr) return fail("TemplateProcessing: unknown special token " + id);
        for (const Json& v : ids->arr) (after ? templateSuffix_ : templatePrefix_).push_back((int32_t)v.i);
      }
    return true;
  } else return fail(std::string(what) + ": component type not supported: " + type);
  out.push_back(std::move(s));
  return true;
}

bool Tokenizer::initWithConfig(const std::string& tokenizerPath, const std::string& cfgPath) {
  bool ok;
  const std::string tj = read_file(tokenizerPath, ok);
  if (!ok) return fail("Cannot open file: " + tokenizerPath);
  Json j;
  if (!JsonParser::parse(tj.data(), tj.size(), j) || j.kind != Json::Obj || j.obj.empty()) return fail("Parse tokenizer error: " + tokenizerPath);
  const std::string cj = read_file(cfgPath, ok);
  if (!ok) return fail("Cannot open file: " + cfgPath);
  Json c;
  if (!JsonParser::parse(cj.data(), cj.size(), c) || c.kind != Json::Obj || c.obj.empty()) return fail("Parse config error: " + cfgPath);

  const Json* model = j.get("model");
  if (!model || model->kind != Json::Obj) return fail("tokenizer.json has no model");
  const std::string mtype = model->get_str("type", "BPE");
  if (mtype != "BPE") return fail("Component type not support: " + mtype);
  const Json* vocab = model->get("vocab");
  if (!vocab || vocab->kind != Json::Obj) return fail("BPE model without vocab");
  int32_t max_id = -1;
  for (const auto& kv : vocab->obj) max_id = std::max(max_id, (int32_t)kv.second.i);
  std::vector<std::pair<std::string, int32_t>> added;
  if (const Json* at = j.get("added_tokens"))
    if (at->kind == Json::Arr)
      for (const Json& t : at->arr) {
        const std::string content = t.get_str("content", "");
        const int32_t id = (int32_t)t.get_int("id", -1);
        if (content.empty() || id < 0) continue;
        max_id = std::max(max_id, id);
        if (content.find("reserved_special_token") != std::string::npos) continue;   // skipped by the reference (Tokenizer.cpp:46-50)
        added.emplace_back(content, id);
      }
  idToToken_.assign((size_t)max_id + 1, std::string());
  isAdded_.assign((size_t)max_id + 1, false);
  vocab_.reserve(vocab->obj.size() * 2);
  for (const auto& kv : vocab->obj) { vocab_[kv.first] = (int32_t)kv.second.i; idToToken_[(size_t)kv.second.i] = kv.first; }
  for (const auto& a : added) { idToToken_[(size_t)a.second] = a.first; isAdded_[(size_t)a.second] = true; }
  added_ = added;
  std::sort(added_.begin(), added_.end(), [](const std::pair<std::string, int32_t>& x, const std::pair<std::string, int32_t>& y) {
    return x.first.size() != y.first.size() ? x.first.size() > y.first.size() : x.first < y.first; });

  ignoreMerges_ = model->get_bool("ignore_merges", false);
  byteFallback_ = model->get_bool("byte_fallback", false);
  { const std::string unk = model->get_str("unk_token", ""); auto it = vocab_.find(unk); unkId_ = (unk.empty() || it == vocab_.end()) ? -1 : it->second; }
  for (int b = 0; b < 256; b++) {
    char buf[8];
    snprintf(buf, sizeof buf, "<0x%02X>", b);
    auto it = vocab_.find(buf);
    byteTokenId_[b] = it == vocab_.end() ? -1 : it->second;
  }
  if (const Json* merges = model->get("merges"))
    if (merges->kind == Json::Arr) {
      int32_t rank = 0;
      merges_.reserve(merges->arr.size() * 2);
      for (const Json& m : merges->arr) {
        std::string a, b;
        if (m.kind == Json::Str) { const size_t sp = m.str.find(' '); if (sp == std::string::npos) continue; a = m.str.substr(0, sp); b = m.str.substr(sp + 1); }
        else if (m.kind == Json::Arr && m.arr.size() == 2) { a = m.arr[0].str; b = m.arr[1].str; }
        else continue;
        auto ia = vocab_.find(a), ib = vocab_.find(b), iab = vocab_.find(a + b);
        if (ia != vocab_.end() && ib != vocab_.end() && iab != vocab_.end()) merges_.emplace(std::make_pair(ia->second, ib->second), std::make_pair(rank, iab->second));
        rank++;
      }
    }

  byteLevelDecode_ = false;
  normalizer_.clear(); preTokenizer_.clear(); decoder_.clear(); templatePrefix_.clear(); templateSuffix_.clear();
  std::vector<Step> post_unused;
  if (const Json* n = j.get("normalizer")) if (!parseSteps(n, normalizer_, "normalizer")) return false;
  if (const Json* p = j.get("pre_tokenizer")) if (!parseSteps(p, preTokenizer_, "pre_tokenizer")) return false;
  if (const Json* p = j.get("post_processor")) if (!parseSteps(p, post_unused, "post_processor")) return false;
  if (const Json* d = j.get("decoder")) if (!parseSteps(d, decoder_, "decoder")) return false;
  for (const Step& s : normalizer_) if (s.type > Step::NFKD) return fail("normalizer: only NFC / NFD / NFKC / NFKD are supported");

  // tokenizer_config.json (TokenizerConfig.cpp:343-371): a token is either a string or an AddedToken object
  auto token_of = [&](const char* key) -> std::string {
    const Json* t = c.get(key);
    if (!t) return {};
    if (t->kind == Json::Str) return t->str;
    if (t->kind == Json::Obj) return t->get_str("content", "");
    return {};
  };
  addBosToken_ = c.get_bool("add_bos_token", false);
  addEosToken_ = c.get_bool("add_eos_token", false);
  bosTokenId_ = token2Id(token_of("bos_token"));
  eosTokenId_ = token2Id(token_of("eos_token"));
  padTokenId_ = token2Id(token_of("pad_token"));
  if ((addBosToken_ && bosTokenId_ < 0) || (addEosToken_ && eosTokenId_ < 0)) return fail("add_bos_token / add_eos_token without a known bos / eos token");
  return true;
}

int32_t Tokenizer::token2Id(const std::string& token) const {
  if (token.empty()) return -1;
  for (const auto& a : added_) if (a.first == token) return a.second;
  auto it = vocab_.find(token);
  return it == vocab_.end() ? -1 : it->second;
}

std::string Tokenizer::id2Token(int32_t id) const {
  if (id < 0 || (size_t)id >= idToToken_.size()) return {};
  if (isAdded_[(size_t)id] || !byteLevelDecode_) return idToToken_[(size_t)id];
  return unicode_to_bytes(idToToken_[(size_t)id]);          // BPE::BPE decoder_ table (BPE.cpp:72)
}

// leftmost match, longest token first (== the ISOLATED split on the added-token alternation, Tokenizer.cpp:293-304)
std::vector<std::string> Tokenizer::splitAddedTokens(const std::string& text) const {
  std::vector<std::string> out;
  if (added_.empty()) { out.push_back(text); return out; }
  size_t seg = 0, i = 0;
  while (i < text.size()) {
    const std::string* hit = nullptr;
    for (const auto& a : added_)
      if (a.first.size() <= text.size() - i && a.first[0] == text[i] && !text.compare(i, a.first.size(), a.first)) { hit = &a.first; break; }
    if (!hit) { i++; continue; }
    if (i > seg) out.emplace_back(text, seg, i - seg);
    out.push_back(*hit);
    i += hit->size();
    seg = i;
  }
  if (seg < text.size()) out.emplace_back(text, seg, text.size() - seg);
  return out;
}

// Word-level BPE on token ids: lowest merge rank first, leftmost on ties (BPE.cpp:162-335 reaches the same result on strings)
void Tokenizer::bpe(const std::string& piece, std::vector<int32_t>& out) const {
  if (piece.empty()) return;
  if (ignoreMerges_) { auto it = vocab_.find(piece); if (it != vocab_.end()) { out.push_back(it->second); return; } }
  std::vector<int32_t> sym;
  for (size_t i = 0; i < piece.size();) {
    uint32_t cp;
    const size_t n = utf8_decode(piece.data() + i, piece.size() - i, cp);
    auto it = vocab_.find(piece.substr(i, n));
    if (it != vocab_.end()) sym.push_back(it->second);
    else {                                           // unknown character: <0xXX> byte tokens (BPE.cpp:146-160), else unk
      bool bytes_ok = true;
      for (size_t k = 0; k < n; k++) bytes_ok &= byteTokenId_[(unsigned char)piece[i + k]] >= 0;
      if (bytes_ok) for (size_t k = 0; k < n; k++) sym.push_back(byteTokenId_[(unsigned char)piece[i + k]]);
      else if (unkId_ >= 0) { if (sym.empty() || sym.back() != unkId_) sym.push_back(unkId_); }
    }
    i += n;
  }
  const size_t n = sym.size();
  if (n < 2) { out.insert(out.end(), sym.begin(), sym.end()); return; }
  std::vector<int> prev(n), next(n);
  for (size_t i = 0; i < n; i++) { prev[i] = (int)i - 1; next[i] = i + 1 < n ? (int)i + 1 : -1; }
  struct Cand { int32_t rank; int pos; int32_t left, right, merged; };
  auto worse = [](const Cand& a, const Cand& b) { return a.rank != b.rank ? a.rank > b.rank : a.pos > b.pos; };
  std::priority_queue<Cand, std::vector<Cand>, decltype(worse)> heap(worse);
  auto push = [&](int i) {
    if (i < 0 || next[i] < 0) return;
    auto it = merges_.find(std::make_pair(sym[(size_t)i], sym[(size_t)next[i]]));
    if (it != merges_.end()) heap.push({it->second.first, i, sym[(size_t)i], sym[(size_t)next[i]], it->second.second});
  };
  for (size_t i = 0; i + 1 < n; i++) push((int)i);
  while (!heap.empty()) {
    const Cand c = heap.top();
    heap.pop();
    const int i = c.pos, r = sym[(size_t)i] == c.left ? next[i] : -1;     // stale entries: a side was merged away since
    if (sym[(size_t)i] < 0 || r < 0 || sym[(size_t)r] != c.right) continue;
    sym[(size_t)i] = c.merged;
    sym[(size_t)r] = -1;
    next[i] = next[r];
    if (next[r] >= 0) prev[next[r]] = i;
    push(prev[i]);
    push(i);
  }
  for (int i = 0; i >= 0; i = next[i]) out.push_back(sym[(size_t)i]);
}

// normalizer -> pre-tokenizer -> model -> template (Tokenizer.cpp:306-321)
std::vector<int32_t> Tokenizer::encodeWithModel(const std::string& text) const {
  std::string norm = text;
  for (const Step& s : normalizer_) norm = normalize_unicode(norm, s.type == Step::NFC || s.type == Step::NFKC, s.type == Step::NFKC || s.type == Step::NFKD);
  std::vector<std::string> pieces{norm};
  for (const Step& s : preTokenizer_) {
    std::vector<std::string> next;
    switch (s.type) {
      case Step::Split:
        for (const std::string& p : pieces) { std::vector<Range> m; s.regex->matchAll(p, m); split_by_matches(p, m, s.behavior, next); }
        break;
      case Step::ByteLevel: {   // ByteLevel.cpp:163-190
        if (s.flag_a && !pieces.empty() && !pieces[0].empty() && pieces[0][0] != ' ') pieces[0].insert(pieces[0].begin(), ' ');
        for (const std::string& p : pieces) {
          if (s.flag_b) {
            std::vector<Range> m; std::vector<std::string> parts;
            s.regex->matchAll(p, m);
            split_by_matches(p, m, Behavior::Isolated, parts);
            for (const std::string& q : parts) next.push_back(bytes_to_unicode(q));
          } else next.push_back(bytes_to_unicode(p));
        }
        break;
      }
      case Step::Metaspace: {   // Metaspace.cpp:15-75: ' ' -> replacement, prepend per scheme (unconditionally, like the reference);
                                // split = a piece per replacement mark, mark first (`tokenizers` MergedWithNext)
        bool first = true;
        for (const std::string& p : pieces) {
          std::string t;
          if (s.str_b == "always" || (s.str_b == "first" && first)) t = s.str_a;
          first = false;
          for (char ch : p) { if (ch == ' ') t += s.str_a; else t += ch; }
          if (!s.flag_a) { next.push_back(std::move(t)); continue; }
          size_t from = 0, at = s.str_a.size() <= t.size() && !t.compare(0, s.str_a.size(), s.str_a) ? s.str_a.size() : 0;
          while ((at = t.find(s.str_a, at)) != std::string::npos) {     // a new piece starts at every replacement mark
            if (at > from) next.emplace_back(t, from, at - from);
            from = at; at += s.str_a.size();
          }
          if (from < t.size()) next.emplace_back(t, from, t.size() - from);
        }
        break;
      }
      default: next = pieces; break;
    }
    pieces.swap(next);
  }
  std::vector<int32_t> ids;
  for (const std::string& p : pieces) bpe(p, ids);
  return ids;
}

// Tokenizer::encode (Tokenizer.cpp:85-128)
std::vector<int32_t> Tokenizer::encode(const std::string& text, bool allowAddedTokens) const {
  std::vector<int32_t> body;
  if (!allowAddedTokens) body = encodeWithModel(text);
  else
    for (const std::string& piece : splitAddedTokens(text)) {
      int32_t added = -1;
      for (const auto& a : added_) if (a.first == piece) { added = a.second; break; }
      if (added >= 0) { body.push_back(added); continue; }
      const std::vector<int32_t> ids = encodeWithModel(piece);
      body.insert(body.end(), ids.begin(), ids.end());
    }
  std::vector<int32_t> ret;
  if (allowAddedTokens) ret = templatePrefix_;           // the special tokens of the "single" template, once per sequence
  ret.insert(ret.end(), body.begin(), body.end());
  if (allowAddedTokens) ret.insert(ret.end(), templateSuffix_.begin(), templateSuffix_.end());
  const bool insert_bos = addBosToken_ && (ret.empty() || ret.front() != bosTokenId_);
  const bool insert_eos = addEosToken_ && (ret.empty() || ret.back() != eosTokenId_);
  if (insert_bos) ret.insert(ret.begin(), bosTokenId_);
  if (insert_eos) ret.push_back(eosTokenId_);
  return ret;
}

std::vector<std::vector<int32_t>> Tokenizer::encodeBatch(const std::vector<std::string>& texts, uint32_t numThreads, bool allowAddedTokens) const {
  std::vector<std::vector<int32_t>> out(texts.size());
  const uint32_t nt = std::max<uint32_t>(1, std::min<uint32_t>(numThreads, (uint32_t)texts.size()));
  if (nt <= 1) { for (size_t i = 0; i < texts.size(); i++) out[i] = encode(texts[i], allowAddedTokens); return out; }
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < nt; t++)
    pool.emplace_back([&, t]() { for (size_t i = t; i < texts.size(); i += nt) out[i] = encode(texts[i], allowAddedTokens); });
  for (auto& th : pool) th.join();
  return out;
}

std::vector<std::string> Tokenizer::runDecoder(std::vector<std::string> pieces) const {
  for (const Step& s : decoder_) {
    switch (s.type) {
      case Step::Replace:
        if (!s.str_a.empty())
          for (std::string& p : pieces)
            for (size_t at = 0; (at = p.find(s.str_a, at)) != std::string::npos; at += s.str_b.size()) p.replace(at, s.str_a.size(), s.str_b);
        break;
      case Step::Metaspace:     // Metaspace.cpp:77-100
        for (size_t i = 0; i < pieces.size(); i++) {
          std::string& p = pieces[i];
          std::string t;
          for (size_t at = 0; at < p.size();) {
            if (!p.compare(at, s.str_a.size(), s.str_a)) { if (!(i == 0 && at == 0 && s.str_b != "never")) t += ' '; at += s.str_a.size(); }
            else t += p[at++];
          }
          p.swap(t);
        }
        break;
      case Step::ByteFallback: {   // runs of <0xXX> tokens -> their bytes if valid UTF-8, else one U+FFFD per token (ByteFallback.cpp:13-52)
        std::vector<std::string> next;
        std::string run; size_t run_len = 0;
        auto flush = [&]() {
          if (!run_len) return;
          if (valid_utf8(run)) next.push_back(run); else for (size_t k = 0; k < run_len; k++) next.emplace_back("\xEF\xBF\xBD");
          run.clear(); run_len = 0;
        };
        for (const std::string& p : pieces) {
          unsigned v;
          if (p.size() == 6 && !p.compare(0, 3, "<0x") && p[5] == '>' && isxdigit((unsigned char)p[3]) && isxdigit((unsigned char)p[4]) && sscanf(p.c_str() + 3, "%2x", &v) == 1) { run += (char)v; run_len++; }
          else { flush(); next.push_back(p); }
        }
        flush();
        pieces.swap(next);
        break;
      }
      case Step::Fuse: {
        std::string all;
        for (const std::string& p : pieces) all += p;
        pieces.assign(1, all);
        break;
      }
      case Step::Strip:          // Strip.cpp:17-40
        for (std::string& p : pieces) {
          size_t a = 0, b = p.size();
          while (a < p.size() && (int)a < s.start && p[a] == s.str_a[0]) a++;
          while (b > a && (int)(p.size() - b) < s.stop && p[b - 1] == s.str_a[0]) b--;
          p = p.substr(a, b - a);
        }
        break;
      default: break;            // ByteLevel: id2Token already produced raw bytes
    }
  }
  return pieces;
}

std::string Tokenizer::decode(const std::vector<int32_t>& ids, uint32_t offset) const {
  std::vector<std::string> pieces;
  for (size_t i = offset; i < ids.size(); i++) pieces.push_back(id2Token(ids[i]));
  std::string out;
  for (const std::string& p : runDecoder(std::move(pieces))) out += p;
  return out;
}

std::vector<std::string> Tokenizer::decodeBatch(const std::vector<std::vector<int32_t>>& ids, uint32_t) const {
  std::vector<std::string> out;
  for (const auto& v : ids) out.push_back(decode(v, 0));
  return out;
}

std::vector<std::string> Tokenizer::decodeBatch(const std::vector<int32_t>& ids, uint32_t batch, uint32_t offset, uint32_t) const {
  std::vector<std::string> out;
  if (!batch || ids.size() % batch) return out;
  const size_t len = ids.size() / batch;
  for (uint32_t b = 0; b < batch; b++) out.push_back(decode(std::vector<int32_t>(ids.begin() + b * len, ids.begin() + (b + 1) * len), offset));
  return out;
}

// Streaming (Tokenizer.cpp:193-267): the token-local decoder stages run per token (Replace, <0xXX> -> byte; Fuse and
// Strip concern whole sequences and do not apply to a continuation), then only complete UTF-8 leaves the cache.
std::string Tokenizer::decodeStream(const std::vector<int32_t>& ids) {
  for (int32_t id : ids) {
    std::string p = id2Token(id);
    unsigned v;
    bool byte_tok = false;
    for (const Step& s : decoder_) {
      if (s.type == Step::Replace && !s.str_a.empty())
        for (size_t at = 0; (at = p.find(s.str_a, at)) != std::string::npos; at += s.str_b.size()) p.replace(at, s.str_a.size(), s.str_b);
      if (s.type == Step::Metaspace)
        for (size_t at = 0; (at = p.find(s.str_a, at)) != std::string::npos; at += 1) p.replace(at, s.str_a.size(), " ");
      if (s.type == Step::ByteFallback && id >= 0 && (size_t)id < isAdded_.size() && !isAdded_[(size_t)id] && p.size() == 6 && !p.compare(0, 3, "<0x") && p[5] == '>' && sscanf(p.c_str() + 3, "%2x", &v) == 1) byte_tok = true;
    }
    if (byte_tok) streamCache_ += (char)v; else streamCache_ += p;
  }
  const size_t cut = incomplete_utf8_tail(streamCache_);
  if (cut == std::string::npos) { std::string out; out.swap(streamCache_); return out; }
  std::string out = streamCache_.substr(0, cut);
  streamCache_.erase(0, cut);
  return out;
}

std::string Tokenizer::decodeStreamFlush() {
  std::string out;
  out.swap(streamCache_);
  return out;
}

}  // namespace tgxh
