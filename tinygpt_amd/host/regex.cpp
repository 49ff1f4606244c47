// regex.cpp — see regex.h.  Pattern -> syntax tree -> backtracking program (CHAR / CLASS / SPLIT / JMP / LOOK /
// MATCH) run with an explicit stack, so a `\p{L}+` over a 500 000-letter word (the reference's long-text test,
// test_tokenizer.cpp:250-262) costs heap, not recursion depth.
#include "regex.h"

#include <algorithm>
#include <climits>

#include "unicode_tables.h"

namespace tgxh {

// ---------------------------------------------------------------------------------------------- UTF-8 / classes
size_t utf8_decode(const char* s, size_t n, uint32_t& cp) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
  const unsigned char c = p[0];
  if (c < 0x80) { cp = c; return 1; }
  int len = (c >= 0xF0 && c <= 0xF4) ? 4 : (c >= 0xE0) ? 3 : (c >= 0xC2 && c < 0xE0) ? 2 : 0;
  if (len == 0 || (size_t)len > n || c > 0xF4) { cp = 0xFFFD; return 1; }
  uint32_t v = c & (0xFF >> (len + 1));
  for (int k = 1; k < len; k++) {
    if ((p[k] & 0xC0) != 0x80) { cp = 0xFFFD; return 1; }
    v = (v << 6) | (p[k] & 0x3F);
  }
  if ((len == 3 && (v < 0x800 || (v >= 0xD800 && v <= 0xDFFF))) || (len == 4 && (v < 0x10000 || v > 0x10FFFF))) { cp = 0xFFFD; return 1; }
  cp = v;
  return (size_t)len;
}

void utf8_append(std::string& out, uint32_t cp) {
  if (cp < 0x80) out += (char)cp;
  else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
  else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
  else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
}

// general category index (unicode_tables.h kCategoryNames) of a code point
static int category_of(uint32_t cp) {
  int lo = 0, hi = unidata::kCategoryCount - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) / 2;
    if (cp < unidata::kCategory[mid].lo) hi = mid - 1;
    else if (cp > unidata::kCategory[mid].hi) lo = mid + 1;
    else return unidata::kCategory[mid].cat;
  }
  return unidata::kCategoryNameCount - 1;   // Cn
}
// bit mask over the category indices for a \p{..} name: a two-letter category, or a one-letter group ("L" = Lu|Ll|Lt|Lm|Lo)
static uint32_t category_mask(const std::string& name) {
  uint32_t m = 0;
  for (int i = 0; i < unidata::kCategoryNameCount; i++) {
    const char* c = unidata::kCategoryNames[i];
    if ((name.size() == 2 && name[0] == c[0] && name[1] == c[1]) || (name.size() == 1 && name[0] == c[0])) m |= 1u << i;
  }
  if (name == "Letter") return category_mask("L");
  if (name == "Number") return category_mask("N");
  if (name == "Punctuation") return category_mask("P");
  if (name == "Symbol") return category_mask("S");
  if (name == "Mark") return category_mask("M");
  if (name == "Separator") return category_mask("Z");
  return m;
}
static const uint32_t kMaskL = 0x1Fu, kMaskN = 0x700u;      // Lu..Lo = indices 0-4, Nd Nl No = 8-10
bool is_letter(uint32_t cp) { return (kMaskL >> category_of(cp)) & 1u; }
bool is_number(uint32_t cp) { return (kMaskN >> category_of(cp)) & 1u; }
bool is_white_space(uint32_t cp) {   // Unicode White_Space (what `\s` means for the regex engines behind tokenizers)
  return (cp >= 0x9 && cp <= 0xD) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
         cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
static bool is_digit(uint32_t cp) { return cp >= '0' && cp <= '9'; }
static uint32_t fold_ascii(uint32_t cp) { return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp; }

enum { P_S = 4, P_D = 8, P_W = 16 };   // white space, ASCII digit, word; \\p{..} classes live in the category masks

bool Regex::CharClass::matches(uint32_t cp) const {
  auto hit = [&](uint32_t c) {
    for (const ClassItem& it : items) if (c >= it.lo && c <= it.hi) return true;
    const uint32_t pos = props & 31u, neg = (props >> 5) & 31u;
    if (cat_pos || cat_neg || pos || neg) {
      const int cat = category_of(c);
      if ((cat_pos >> cat) & 1u) return true;
      for (uint32_t nm : cat_neg_list) if (!((nm >> cat) & 1u)) return true;   // \P{X}: any code point outside X
      const bool s = is_white_space(c), d = is_digit(c), w = (c == '_') || ((kMaskL | kMaskN) >> cat & 1u);
      if (((pos & P_S) && s) || ((pos & P_D) && d) || ((pos & P_W) && w)) return true;
      if (((neg & P_S) && !s) || ((neg & P_D) && !d) || ((neg & P_W) && !w)) return true;
    }
    return false;
  };
  bool m = hit(cp);
  if (!m && icase) {
    if (cp >= 'a' && cp <= 'z') m = hit(cp - 32);
    else if (cp >= 'A' && cp <= 'Z') m = hit(cp + 32);
  }
  return m != negated;
}

// ---------------------------------------------------------------------------------------------- parser
namespace {
struct Node {
  enum Kind { Char, Any, Class, Cat, Alt, Repeat, Look, Bol, Eol } kind = Cat;
  uint32_t cp = 0;
  bool icase = false;
  int cls = -1;
  int min = 0, max = 0;      // Repeat; max < 0 = unbounded
  bool lazy = false, neg = false;
  std::vector<Node> kids;
};
}  // namespace

struct Regex::Parser {
  Regex& re;
  std::vector<uint32_t> p;   // pattern code points
  size_t i = 0;
  std::string err;
  explicit Parser(Regex& r, const std::string& pat) : re(r) {
    for (size_t k = 0; k < pat.size();) { uint32_t cp; k += utf8_decode(pat.data() + k, pat.size() - k, cp); p.push_back(cp); }
  }
  bool more() const { return i < p.size(); }
  bool fail(const char* m) { if (err.empty()) err = m; return false; }

  bool alt(Node& out, bool icase) {
    Node first;
    if (!cat(first, icase)) return false;
    if (!(more() && p[i] == '|')) { out = std::move(first); return true; }
    out = Node(); out.kind = Node::Alt;
    out.kids.push_back(std::move(first));
    while (more() && p[i] == '|') {
      i++;
      Node next;
      if (!cat(next, icase)) return false;
      out.kids.push_back(std::move(next));
    }
    return true;
  }
  static bool nullable(const Node& n) {
    switch (n.kind) {
      case Node::Char: case Node::Any: case Node::Class: return false;
      case Node::Bol: case Node::Eol: case Node::Look: return true;
      case Node::Cat: for (const Node& k : n.kids) if (!nullable(k)) return false; return true;
      case Node::Alt: for (const Node& k : n.kids) if (nullable(k)) return true; return false;
      case Node::Repeat: return n.min == 0 || nullable(n.kids[0]);
    }
    return true;
  }
  bool cat(Node& out, bool icase) {
    out = Node(); out.kind = Node::Cat;
    while (more() && p[i] != '|' && p[i] != ')') {
      Node a;
      if (!atom(a, icase)) return false;
      while (more() && (p[i] == '?' || p[i] == '*' || p[i] == '+' || p[i] == '{')) {
        int mn, mx;
        if (p[i] == '?') { mn = 0; mx = 1; i++; }
        else if (p[i] == '*') { mn = 0; mx = -1; i++; }
        else if (p[i] == '+') { mn = 1; mx = -1; i++; }
        else {
          size_t save = i++;
          auto num = [&](int& v) { bool any = false; v = 0; while (more() && p[i] >= '0' && p[i] <= '9') { v = v * 10 + (int)(p[i++] - '0'); any = true; if (v > 1000) return false; } return any; };
          if (!num(mn)) { i = save; break; }          // a literal '{'
          mx = mn;
          if (more() && p[i] == ',') { i++; if (more() && p[i] == '}') mx = -1; else if (!num(mx)) return fail("bad {n,m}"); }
          if (!(more() && p[i] == '}')) return fail("unterminated {n,m}");
          i++;
          if (mx >= 0 && mx < mn) return fail("{n,m} with m < n");
        }
        // an unbounded repeat of something that can match the empty string never terminates in a backtracking matcher (and no
        // pre-tokenizer pattern needs one): refused at load
        if (mx < 0 && nullable(a)) return fail("unbounded repeat of an expression that can match the empty string");
        Node r; r.kind = Node::Repeat; r.min = mn; r.max = mx;
        if (more() && p[i] == '?') { r.lazy = true; i++; }
        else if (more() && p[i] == '+') return fail("possessive quantifiers are not supported");
        r.kids.push_back(std::move(a));
        a = std::move(r);
      }
      out.kids.push_back(std::move(a));
    }
    return true;
  }
  // after \p or \P: {Name} or a single letter; Name = a Unicode general category (Lu, Nd, ...) or a one-letter group (L, N, P, S, M, Z, C)
  bool prop(CharClass& cc, bool negate) {
    std::string name;
    if (more() && p[i] == '{') { i++; while (more() && p[i] != '}') name += (char)p[i++]; if (!more()) return fail("unterminated \\p{"); i++; }
    else if (more()) name += (char)p[i++];
    const uint32_t mask = category_mask(name);
    if (!mask) return fail("unsupported \\p{..} class (Unicode general categories are built; scripts and binary properties are not)");
    if (negate) { cc.cat_neg |= mask; cc.cat_neg_list.push_back(mask); } else cc.cat_pos |= mask;
    return true;
  }
  // one escape sequence: either a literal code point (lit) or class property flags
  bool escape(uint32_t& lit, CharClass& cc, bool& is_class) {
    if (!more()) return fail("trailing backslash");
    const uint32_t e = p[i++];
    uint32_t& flags = cc.props;
    is_class = true;
    switch (e) {
      case 's': flags |= P_S; return true;          case 'S': flags |= P_S << 5; return true;
      case 'd': flags |= P_D; return true;          case 'D': flags |= P_D << 5; return true;
      case 'w': flags |= P_W; return true;          case 'W': flags |= P_W << 5; return true;
      case 'p': return prop(cc, false);             case 'P': return prop(cc, true);
      default: break;
    }
    is_class = false;
    switch (e) {
      case 'n': lit = '\n'; return true;  case 'r': lit = '\r'; return true;  case 't': lit = '\t'; return true;
      case 'f': lit = '\f'; return true;  case 'v': lit = '\v'; return true;  case '0': lit = 0; return true;
      case 'x': case 'u': {
        uint32_t v = 0; int digits = 0;
        if (more() && p[i] == '{') { i++; while (more() && p[i] != '}') { v = v * 16 + hexval(p[i++]); digits++; } if (!more()) return fail("unterminated \\x{"); i++; }
        else { const int want = e == 'x' ? 2 : 4; while (digits < want && more() && hexval(p[i]) < 16) { v = v * 16 + hexval(p[i++]); digits++; } }
        if (!digits) return fail("bad hex escape");
        lit = v; return true;
      }
      default:
        if ((e >= 'a' && e <= 'z') || (e >= 'A' && e <= 'Z') || (e >= '1' && e <= '9')) return fail("unsupported escape");
        lit = e; return true;
    }
  }
  static uint32_t hexval(uint32_t c) { return (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : 99; }

  bool char_class(Node& out, bool icase) {   // after '['
    CharClass cc; cc.icase = icase;
    if (more() && p[i] == '^') { cc.negated = true; i++; }
    bool first = true;
    while (more() && (p[i] != ']' || first)) {
      first = false;
      uint32_t lo; bool is_class = false;
      if (p[i] == '\\') { i++; if (!escape(lo, cc, is_class)) return false; }
      else lo = p[i++];
      if (is_class) continue;
      uint32_t hi = lo;
      if (i + 1 < p.size() && p[i] == '-' && p[i + 1] != ']') {
        i++;
        bool c2 = false; CharClass dummy;
        if (p[i] == '\\') { i++; if (!escape(hi, dummy, c2)) return false; if (c2) return fail("class escape as range end"); }
        else hi = p[i++];
        if (hi < lo) return fail("reversed range");
      }
      cc.items.push_back({lo, hi});
    }
    if (!more()) return fail("unterminated [");
    i++;
    out = Node(); out.kind = Node::Class; out.cls = (int)re.classes_.size();
    re.classes_.push_back(std::move(cc));
    return true;
  }
  bool atom(Node& out, bool icase) {
    const uint32_t c = p[i++];
    out = Node();
    if (c == '(') {
      bool ic = icase; int look = 0;   // 1 positive, 2 negative
      if (more() && p[i] == '?') {
        i++;
        if (more() && p[i] == ':') i++;
        else if (more() && p[i] == '=') { look = 1; i++; }
        else if (more() && p[i] == '!') { look = 2; i++; }
        else if (more() && p[i] == 'i' && i + 1 < p.size() && p[i + 1] == ':') { ic = true; i += 2; }
        else return fail("unsupported group flag");
      }
      Node inner;
      if (!alt(inner, ic)) return false;
      if (!(more() && p[i] == ')')) return fail("unterminated (");
      i++;
      if (look) { out.kind = Node::Look; out.neg = look == 2; out.kids.push_back(std::move(inner)); }
      else out = std::move(inner);
      return true;
    }
    if (c == '[') return char_class(out, icase);
    if (c == '.') { out.kind = Node::Any; return true; }
    if (c == '^') { out.kind = Node::Bol; return true; }
    if (c == '$') { out.kind = Node::Eol; return true; }
    if (c == '*' || c == '+' || c == '?') return fail("quantifier without operand");
    if (c == '\\') {
      uint32_t lit = 0; bool is_class = false;
      CharClass cc;
      if (!escape(lit, cc, is_class)) return false;
      if (is_class) {
        cc.icase = icase;
        out.kind = Node::Class; out.cls = (int)re.classes_.size();
        re.classes_.push_back(std::move(cc));
        return true;
      }
      out.kind = Node::Char; out.cp = lit; out.icase = icase;
      return true;
    }
    out.kind = Node::Char; out.cp = c; out.icase = icase;
    return true;
  }

  // ---- code generation
  void emit(const Node& n) {
    std::vector<Inst>& g = re.prog_;
    switch (n.kind) {
      case Node::Char: { Inst in; in.op = CHAR; in.a = n.icase ? fold_ascii(n.cp) : n.cp; in.flag = n.icase; g.push_back(in); break; }
      case Node::Any: { Inst in; in.op = ANY; g.push_back(in); break; }
      case Node::Class: { Inst in; in.op = CLASS; in.a = (uint32_t)n.cls; g.push_back(in); break; }
      case Node::Bol: { Inst in; in.op = BOL; g.push_back(in); break; }
      case Node::Eol: { Inst in; in.op = EOL; g.push_back(in); break; }
      case Node::Cat: for (const Node& k : n.kids) emit(k); break;
      case Node::Alt: {
        std::vector<size_t> jumps;
        for (size_t k = 0; k < n.kids.size(); k++) {
          if (k + 1 < n.kids.size()) {
            const size_t sp = g.size();
            Inst in; in.op = SPLIT; g.push_back(in);
            g[sp].a = (uint32_t)g.size();
            emit(n.kids[k]);
            jumps.push_back(g.size());
            Inst j; j.op = JMP; g.push_back(j);
            g[sp].b = (uint32_t)g.size();
          } else emit(n.kids[k]);
        }
        for (size_t j : jumps) g[j].a = (uint32_t)g.size();
        break;
      }
      case Node::Repeat: {
        const Node& body = n.kids[0];
        for (int k = 0; k < n.min; k++) emit(body);
        if (n.max < 0) {
          const size_t sp = g.size();
          Inst in; in.op = SPLIT; g.push_back(in);
          const size_t b0 = g.size();
          emit(body);
          Inst j; j.op = JMP; j.a = (uint32_t)sp; g.push_back(j);
          const size_t out = g.size();
          g[sp].a = (uint32_t)(n.lazy ? out : b0); g[sp].b = (uint32_t)(n.lazy ? b0 : out);
        } else {
          std::vector<size_t> splits;
          for (int k = n.min; k < n.max; k++) {
            splits.push_back(g.size());
            Inst in; in.op = SPLIT; g.push_back(in);
            g[splits.back()].a = (uint32_t)g.size();   // patched below for lazy
            emit(body);
          }
          const size_t out = g.size();
          for (size_t sp : splits) {
            const uint32_t b0 = (uint32_t)(sp + 1);
            g[sp].a = n.lazy ? (uint32_t)out : b0; g[sp].b = n.lazy ? b0 : (uint32_t)out;
          }
        }
        break;
      }
      case Node::Look: {
        const size_t lk = g.size();
        Inst in; in.op = LOOK; in.flag = n.neg; g.push_back(in);
        g[lk].a = (uint32_t)g.size();
        emit(n.kids[0]);
        Inst m; m.op = MATCH; g.push_back(m);
        g[lk].b = (uint32_t)g.size();
        break;
      }
    }
  }
};

Regex::Regex(const std::string& pattern) {
  Parser ps(*this, pattern);
  Node root;
  if (!ps.alt(root, false)) { err_ = ps.err.empty() ? "syntax error" : ps.err; return; }
  if (ps.more()) { err_ = "unbalanced )"; return; }
  ps.emit(root);
  Inst m; m.op = MATCH; prog_.push_back(m);
}

// ---------------------------------------------------------------------------------------------- matcher
long Regex::run(size_t pc0, const std::vector<uint32_t>& cps, size_t i0, size_t& budget) const {
  struct Thread { uint32_t pc; size_t i; };
  std::vector<Thread> stack;
  stack.push_back({(uint32_t)pc0, i0});
  const size_t n = cps.size();
  while (!stack.empty()) {
    Thread t = stack.back();
    stack.pop_back();
    for (;;) {
      if (budget == 0) return -2;                   // step budget of this matchAll call spent (exponential backtracking): give up
      budget--;
      const Inst& in = prog_[t.pc];
      bool ok = true;
      switch (in.op) {
        case CHAR: ok = t.i < n && (in.flag ? fold_ascii(cps[t.i]) == in.a : cps[t.i] == in.a); if (ok) { t.i++; t.pc++; } break;
        case ANY: ok = t.i < n && cps[t.i] != '\n'; if (ok) { t.i++; t.pc++; } break;
        case CLASS: ok = t.i < n && classes_[in.a].matches(cps[t.i]); if (ok) { t.i++; t.pc++; } break;
        case BOL: ok = t.i == 0; if (ok) t.pc++; break;
        case EOL: ok = t.i == n; if (ok) t.pc++; break;
        case SPLIT: stack.push_back({in.b, t.i}); t.pc = in.a; break;
        case JMP: t.pc = in.a; break;
        case LOOK: { const long r = run(in.a, cps, t.i, budget); if (r == -2) return -2; ok = (r >= 0) != in.flag; if (ok) t.pc = in.b; break; }
        case MATCH: return (long)t.i;
      }
      if (!ok) break;
    }
  }
  return -1;
}

void Regex::matchAll(const std::string& text, std::vector<Range>& out) const {
  if (!valid()) return;
  std::vector<uint32_t> cps;
  std::vector<size_t> off;
  cps.reserve(text.size()); off.reserve(text.size() + 1);
  for (size_t k = 0; k < text.size();) { uint32_t cp; off.push_back(k); k += utf8_decode(text.data() + k, text.size() - k, cp); cps.push_back(cp); }
  off.push_back(text.size());
  // a tokenizer pattern needs a few steps per code point; the budget bounds pathological (exponentially backtracking) patterns: when it
  // is spent, matching stops and the rest of the text stays unmatched (the Split pre-tokenizer still emits it: a valid, coarser split)
  size_t budget = 20000000 + 4000 * cps.size();
  size_t i = 0;
  while (i < cps.size()) {
    const long e = run(0, cps, i, budget);
    if (e == -2) break;
    if (e > (long)i) { out.emplace_back(off[i], off[(size_t)e]); i = (size_t)e; }
    else i++;
  }
}

}  // namespace tgxh
