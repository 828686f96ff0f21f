// contextrules.cpp -- context rules of search mode (the part of most_likely_sequence
// that src/lib.rs:2345-2363 adds): pattern parsing (/root/reference/src/search.rs:413-459), rule files
// (src/lib.rs:570-765), rule matching with tags (src/search.rs:472-524) and the sequence score (src/lib.rs:2501-2578).
// Host code; pinned by the reference's tests/main.rs:1575-1800 (test0902-0905).
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "host_model.h"

namespace anx {

namespace {

std::vector<std::string> split_on(const std::string& s, char sep) {  // str::split: always at least one field
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    const size_t e = s.find(sep, b);
    if (e == std::string::npos) { out.push_back(s.substr(b)); break; }
    out.push_back(s.substr(b, e - b));
    b = e + 1;
  }
  return out;
}

bool ends_with(const std::string& s, const std::string& tail) {
  return s.size() >= tail.size() && s.compare(s.size() - tail.size(), tail.size(), tail) == 0;
}

bool parse_u8(const std::string& s, uint8_t* out) {  // str::parse::<u8>: optional '+', digits only, <= 255
  size_t i = 0;
  if (i < s.size() && s[i] == '+') ++i;
  if (i >= s.size()) return false;
  unsigned v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (unsigned)(s[i] - '0');
    if (v > 255) return false;
  }
  *out = (uint8_t)v;
  return true;
}

bool parse_f32(const std::string& s, float* out) {  // str::parse::<f32>: no surrounding whitespace, all consumed
  if (s.empty() || s[0] == ' ' || s[0] == '\t' || (s.size() > 1 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X'))) return false;
  char* end = nullptr;
  errno = 0;
  const float v = strtof(s.c_str(), &end);
  if (end != s.c_str() + s.size()) return false;
  *out = v;
  return true;
}

// PatternMatch::parse (src/search.rs:413-459)
bool parse_pattern(const std::string& raw, const HostModel& m, PatternMatch& out, std::string& err) {
  const std::string s = trim_whitespace(raw);
  if (s == "?") { out.kind = PatternMatch::Any; return true; }
  if (s == "^") { out.kind = PatternMatch::NoLexicon; return true; }
  if (s.size() >= 3 && s.compare(0, 2, "!(") == 0 && s.back() == ')') {  // negation over a disjunction
    PatternMatch inner;
    if (!parse_pattern(s.substr(2, s.size() - 3), m, inner, err)) return false;
    out.kind = PatternMatch::Not;
    out.sub.assign(1, inner);
    return true;
  }
  if (s.find('|') != std::string::npos) {
    out.kind = PatternMatch::Disjunction;
    for (const std::string& item : split_on(s, '|')) {
      PatternMatch pm;
      if (!parse_pattern(item, m, pm, err)) return false;
      out.sub.push_back(pm);
    }
    return true;
  }
  if (!s.empty() && s[0] == '!') {
    PatternMatch inner;
    if (!parse_pattern(s.substr(1), m, inner, err)) return false;
    out.kind = PatternMatch::Not;
    out.sub.assign(1, inner);
    return true;
  }
  if (!s.empty() && s[0] == '@') {
    const std::string source = s.substr(1), rel = "/" + source;
    for (size_t i = 0; i < m.lexicons.size(); ++i)
      if (source == m.lexicons[i] || ends_with(m.lexicons[i], rel)) {
        out.kind = PatternMatch::FromLexicon;
        out.lexicon = (uint8_t)i;
        return true;
      }
    err = "WARNING: Context rule references lexicon or variant list '" + source + "' but this source was not loaded";
    return false;
  }
  auto it = m.encoder.find(s);
  if (it != m.encoder.end()) {
    out.kind = PatternMatch::Vocab;
    out.vocab_id = it->second;
    return true;
  }
  err = "WARNING: Context rule references word '" + s + "' but this word does not occur in any lexicon";
  return false;
}

}  // namespace

bool PatternMatch::matches(uint64_t id, uint32_t lexindex) const {
  switch (kind) {
    case Any: return true;
    case NoLexicon: return lexindex == 0 || id == 0;
    case Vocab: return id == vocab_id;
    case FromLexicon: return lexicon < 32 && (lexindex >> lexicon) & 1u;
    case Not: return !sub[0].matches(id, lexindex);
    case Disjunction:
      for (const PatternMatch& pm : sub)
        if (pm.matches(id, lexindex)) return true;
      return false;
  }
  return false;
}

int HostModel::add_contextrule(const std::string& pattern_s, float score, const std::vector<std::string>& tag_s,
                               const std::vector<std::string>& tagoffset_s, std::string& err) {
  std::vector<PatternMatch> pattern;
  for (const std::string& expr : split_on(pattern_s, ';')) {
    PatternMatch pm;
    std::string perr;
    if (!parse_pattern(expr, *this, pm, perr)) { err = "Error parsing context rule: " + perr; return ANX_EINVAL; }
    pattern.push_back(pm);
  }
  // tags are interned before the checks, like the reference does (src/lib.rs:680-700)
  bool empty_tag = false;
  std::vector<uint16_t> tag;
  for (const std::string& t : tag_s) {
    if (t.empty()) empty_tag = true;
    size_t pos = tags.size();
    for (size_t i = 0; i < tags.size(); ++i)
      if (tags[i] == t) { pos = i; break; }
    if (pos == tags.size()) tags.push_back(t);
    tag.push_back((uint16_t)pos);
  }
  if (empty_tag) { err = "tag is empty"; return ANX_EINVAL; }
  const char* oerr = nullptr;
  std::vector<std::pair<uint8_t, uint8_t>> tagoffset;
  for (const std::string& s : tagoffset_s) {
    const std::vector<std::string> fields = split_on(s, ':');
    uint8_t begin = 0, length;
    if (!fields[0].empty() && !parse_u8(fields[0], &begin)) { oerr = "tag offset should be an integer"; begin = 0; }
    if (fields.size() > 1 && !fields[1].empty()) {
      if (!parse_u8(fields[1], &length)) { oerr = "tag length should be an integer"; length = 0; }
    } else {
      length = (uint8_t)(pattern.size() - begin);
    }
    tagoffset.emplace_back(begin, length);
  }
  if (oerr) { err = oerr; return ANX_EINVAL; }
  while (tagoffset.size() < tag.size()) tagoffset.emplace_back((uint8_t)0, (uint8_t)pattern.size());
  if (!pattern.empty()) {
    ContextRule r;
    r.pattern = std::move(pattern);
    r.score = score;
    r.tag = std::move(tag);
    r.tagoffset = std::move(tagoffset);
    context_rules.push_back(std::move(r));
  }
  return ANX_OK;
}

int HostModel::read_contextrules(const std::string& path, std::string& err) {
  std::ifstream f(path);
  if (!f) { err = "cannot open " + path; return ANX_EIO; }
  std::string line;
  size_t linenr = 0;
  auto where = [&]() { return " (" + path + ", line " + std::to_string(linenr) + ")"; };
  auto tag_fields = [](const std::string& s) {
    std::vector<std::string> out;
    for (const std::string& w : split_on(s, ';')) {
      const std::string t = trim_whitespace(w);
      if (!t.empty()) out.push_back(t);
    }
    return out;
  };
  while (std::getline(f, line)) {
    ++linenr;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    const std::vector<std::string> fields = split_on(line, '\t');
    if (fields.size() < 2) {
      err = "Expected at least two columns in context rules file " + path + ", line " + std::to_string(linenr);
      return ANX_EINVAL;
    }
    if (fields[0].empty()) continue;
    float score;
    if (!parse_f32(fields[1], &score)) {
      err = "context rule score should be a floating point value above or below 1.0, got " + fields[1] + where();
      return ANX_EINVAL;
    }
    std::vector<std::string> tag = fields.size() > 2 ? tag_fields(fields[2]) : std::vector<std::string>();
    std::vector<std::string> tagoffset = fields.size() > 3 ? tag_fields(fields[3]) : std::vector<std::string>();
    if (tag.size() == 1 && tagoffset.empty()) tagoffset.push_back("0:");
    else if (tag.size() != tagoffset.size()) {
      err = "Multiple tags are specified for a context rule, expected the same number of tag offsets! (semicolon separated)" + where();
      return ANX_EINVAL;
    }
    std::string aerr;
    if (add_contextrule(fields[0], score, tag, tagoffset, aerr) != ANX_OK) {
      err = "Error adding context rule: " + aerr + where();
      return ANX_EINVAL;
    }
  }
  return ANX_OK;
}

double HostModel::test_context_rules(const std::vector<std::pair<uint64_t, uint32_t>>& seq,
                                     std::vector<std::vector<PatternMatchResult>>& results) const {
  results.assign(seq.size(), {});
  bool found = false;
  for (size_t begin = 0; begin < seq.size(); ++begin)
    for (const ContextRule& rule : context_rules) {
      // ContextRule::matches (src/search.rs:472-524): all positions still uncovered and matching
      const size_t len = rule.pattern.size();
      if (begin + len > seq.size()) continue;
      bool ok = true;
      for (size_t c = 0; c < len && ok; ++c)
        ok = results[begin + c].empty() && rule.pattern[c].matches(seq[begin + c].first, seq[begin + c].second);
      if (!ok) continue;
      found = true;
      for (size_t c = 0; c < len; ++c) {
        std::vector<PatternMatchResult>& dst = results[begin + c];
        dst.clear();
        if (rule.tag.empty()) dst.push_back(PatternMatchResult{rule.score, -1, (uint8_t)c});
        else
          for (size_t t = 0; t < rule.tag.size() && t < rule.tagoffset.size(); ++t) {
            const unsigned b = rule.tagoffset[t].first, l = rule.tagoffset[t].second;
            if (c >= b && c < b + l) dst.push_back(PatternMatchResult{rule.score, (int32_t)rule.tag[t], (uint8_t)(c - b)});
          }
      }
    }
  if (!found) return 1.0;
  float sum = 0.0f;  // .sum::<f32>()
  for (const auto& r : results) sum += r.empty() ? 1.0f : r[0].score;
  return (double)sum / (double)seq.size();
}

}  // namespace anx
