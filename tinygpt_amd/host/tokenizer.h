// tokenizer.h — HF `tokenizer.json` BPE tokenizer for the engine's text entry points: the counterpart of
// tinygpt::tokenizer::Tokenizer (src/tokenizer/Tokenizer.h:27-120) with the same method names and the same
// observable behaviour on the reference's golden vectors (test/test_tokenizer.cpp:86-262).
//
// Supported tokenizer.json components (the set the reference builds, TokenizerConfig.cpp:27-40):
//   normalizer     NFC / NFD / NFKC / NFKD (generated Unicode tables), Sequence
//   pre_tokenizer  ByteLevel, Split (regex or string pattern, all five delimiter behaviours), Metaspace, Sequence
//   model          BPE (vocab + merges, ignore_merges, byte_fallback <0xXX> tokens)
//   post_processor TemplateProcessing (single sequence), ByteLevel (no-op), Sequence
//   decoder        ByteLevel, Metaspace, Replace (string pattern), ByteFallback, Fuse, Strip, Sequence
// Where the reference and the `tokenizers` library disagree this follows `tokenizers` (the golden vectors of the
// reference's own tests are `tokenizers` outputs); the three known spots are listed in DESIGN.md §11.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "regex.h"

namespace tgxh {

// Unicode normalisation forms (UnicodeNorm.h: NFC / NFD / NFKC / NFKD components)
std::string normalize_unicode(const std::string& text, bool compose, bool compat);

class Tokenizer {
 public:
  Tokenizer();
  ~Tokenizer();
  Tokenizer(const Tokenizer&) = delete;
  Tokenizer& operator=(const Tokenizer&) = delete;

  // tokenizer.json + tokenizer_config.json (Tokenizer.cpp:28-66); false + lastError() on failure
  bool initWithConfig(const std::string& tokenizerPath, const std::string& cfgPath);
  const std::string& lastError() const { return err_; }

  int32_t token2Id(const std::string& token) const;
  std::string id2Token(int32_t id) const;        // the token's text as it appears in decoded output (byte-level tokens: raw bytes)

  std::vector<int32_t> encode(const std::string& text, bool allowAddedTokens = true) const;
  std::vector<std::vector<int32_t>> encodeBatch(const std::vector<std::string>& texts, uint32_t numThreads = 8, bool allowAddedTokens = true) const;
  std::string decode(const std::vector<int32_t>& ids, uint32_t offset = 0) const;
  std::vector<std::string> decodeBatch(const std::vector<std::vector<int32_t>>& ids, uint32_t numThreads = 8) const;
  std::vector<std::string> decodeBatch(const std::vector<int32_t>& ids, uint32_t batch, uint32_t offset = 0, uint32_t numThreads = 8) const;

  // streaming: returns only complete UTF-8; an incomplete tail is held back until the next call (Tokenizer.cpp:193-260)
  std::string decodeStream(const std::vector<int32_t>& ids);
  std::string decodeStreamFlush();

  int32_t bosTokenId() const { return bosTokenId_; }
  int32_t eosTokenId() const { return eosTokenId_; }
  int32_t padTokenId() const { return padTokenId_; }
  std::string bosTokenStr() const { return bosTokenId_ < 0 ? std::string() : id2Token(bosTokenId_); }
  std::string eosTokenStr() const { return eosTokenId_ < 0 ? std::string() : id2Token(eosTokenId_); }
  std::string padTokenStr() const { return padTokenId_ < 0 ? std::string() : id2Token(padTokenId_); }
  size_t vocabSize() const { return idToToken_.size(); }

 private:
  struct Step;                                   // one normalizer / pre-tokenizer / decoder stage
  struct PairHash { size_t operator()(const std::pair<int32_t, int32_t>& p) const { return (size_t)p.first * 1000003u ^ (size_t)p.second; } };
  bool fail(const std::string& m) { err_ = m; return false; }
  bool parseSteps(const void* json, std::vector<Step>& out, const char* what);
  std::vector<int32_t> encodeWithModel(const std::string& text) const;
  void bpe(const std::string& piece, std::vector<int32_t>& out) const;
  std::vector<std::string> splitAddedTokens(const std::string& text) const;
  std::vector<std::string> runDecoder(std::vector<std::string> pieces) const;
  std::string decodeRaw(const std::vector<int32_t>& ids, size_t begin) const;

  std::vector<Step> normalizer_, preTokenizer_, decoder_;
  std::vector<int32_t> templatePrefix_, templateSuffix_;   // TemplateProcessing "single": specials before / after sequence A
  std::unordered_map<std::string, int32_t> vocab_;
  std::vector<std::string> idToToken_;           // model vocab + added tokens, as written in tokenizer.json
  std::vector<bool> isAdded_;
  std::unordered_map<std::pair<int32_t, int32_t>, std::pair<int32_t, int32_t>, PairHash> merges_;   // (left id, right id) -> (rank, merged id)
  bool ignoreMerges_ = false, byteFallback_ = false, byteLevelDecode_ = false;
  int32_t unkId_ = -1;
  int32_t byteTokenId_[256];
  std::vector<std::pair<std::string, int32_t>> added_;   // sorted longest first
  int32_t bosTokenId_ = -1, eosTokenId_ = -1, padTokenId_ = -1;
  bool addBosToken_ = false, addEosToken_ = false;
  std::string streamCache_;                      // decoded bytes not yet handed out by decodeStream
  std::vector<int32_t> streamIds_;
  size_t streamEmitted_ = 0;
  std::string err_;
};

}  // namespace tgxh
