// regex.h — a small backtracking regular-expression matcher over Unicode code points, sized for the
// pre-tokenizer patterns HF tokenizer.json files carry (GPT-2, Llama-3, Qwen2/3 `Split` patterns):
// alternation, groups `( )`, `(?: )`, `(?i: )`, look-ahead `(?= )` / `(?! )`, classes with ranges and
// negation, `\p{..}` for every Unicode general category or one-letter group (`\p{L}` `\p{Lu}` `\p{N}` `\p{P}` `\p{S}` `\p{M}` ...),
// `\s` `\d` `\w` (and their negations), quantifiers `? * + {n} {n,} {n,m}`
// (greedy or lazy).  Perl semantics: leftmost match, first alternative that leads to a match.
// The reference wraps PCRE2 for the same job (src/tokenizer/Regex.cpp); only the behaviour at the
// `matchAll` boundary is reproduced here, validated against the `tokenizers` library (tests/golden/tokenizer).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace tgxh {

typedef std::pair<size_t, size_t> Range;   // [first, second) byte offsets

// UTF-8 helpers shared with the tokenizer
size_t utf8_decode(const char* s, size_t n, uint32_t& cp);   // bytes consumed (>= 1; invalid byte -> U+FFFD, 1 byte)
void utf8_append(std::string& out, uint32_t cp);
bool is_letter(uint32_t cp);
bool is_number(uint32_t cp);
bool is_white_space(uint32_t cp);

class Regex {
 public:
  explicit Regex(const std::string& pattern);
  bool valid() const { return err_.empty(); }
  const std::string& error() const { return err_; }
  // every non-overlapping match, scanning left to right (== Regex::matchAll, Regex.h:22); empty matches are skipped
  void matchAll(const std::string& text, std::vector<Range>& out) const;

 private:
  struct ClassItem { uint32_t lo, hi; };
  struct CharClass {
    std::vector<ClassItem> items;
    bool negated = false, icase = false;
    // \s \d \w flags (bit2 White_Space, bit3 ASCII digit, bit4 word; the next 5 bits are their negations)
    uint32_t props = 0;
    // \p{..}: bit mask over the general-category indices of unicode_tables.h; every \P{..} keeps its own mask
    uint32_t cat_pos = 0, cat_neg = 0;
    std::vector<uint32_t> cat_neg_list;
    bool matches(uint32_t cp) const;
  };
  enum Op { CHAR, ANY, CLASS, SPLIT, JMP, LOOK, MATCH, BOL, EOL };
  struct Inst { Op op; uint32_t a = 0, b = 0; bool flag = false; };   // CHAR: a=cp, flag=icase; CLASS: a=index; SPLIT: a first, b second;
                                                                      // LOOK: a = first inst of the sub-program, b = inst after it, flag = negative
  struct Parser;
  std::vector<Inst> prog_;
  std::vector<CharClass> classes_;
  std::string err_;
  // runs the program from `pc` at code point index `i`; returns the end index of the match or -1
  // -1: no match here; -2: the call's step budget is spent
  long run(size_t pc, const std::vector<uint32_t>& cps, size_t i, size_t& budget) const;
};

}  // namespace tgxh
