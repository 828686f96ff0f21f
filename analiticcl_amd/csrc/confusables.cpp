// confusables.cpp -- confusable weighting of ranked results, host side.
// Follows src/confusables.rs:5-128 (pattern syntax, found_in), src/lib.rs:409-458 (loaders), :1505-1508 / :1591-1595
// (early / late rescoring) and :1733-1756 (compute_confusable_weight).  The reference obtains the edit script from
// sesdiff::shortest_edit_script(input, candidate, false, false, false) (src/lib.rs:1736; sesdiff 0.3.1, Cargo.toml:26),
// which maps dissimilar::diff chunks 1:1 to Identity / Deletion / Insertion.  Neither crate is in the reference tree:
// the diff below restates the published algorithm (dissimilar = the Diff part of Google's diff-match-patch: common
// prefix/suffix, containment speed-up, Myers bisect, cleanup_semantic incl. lossless shifts and overlap extraction,
// cleanup_merge; on Unicode scalar values).  PARITY UNPINNED beyond tests/main.rs:914-1020.
#include <algorithm>
#include <cstring>
#include <fstream>

#include "host_model.h"

namespace anx {
namespace {

typedef std::u32string U;  // long-lived text: patterns, the decoded vocabulary

// The diff works on strings of a few dozen code points and builds / splices many short ones per call: with the default
// allocator a ranked row cost ~1 us of malloc / free (u32string keeps 3 characters inline).  Everything a call to edit_script
// allocates comes from a per-thread arena instead (pointer bump, nothing freed, reset at the start of the next call): the
// result of a call is valid until the same thread calls edit_script again.
struct Arena {
  std::vector<std::unique_ptr<char[]>> chunks;
  std::vector<size_t> sizes;
  size_t chunk = 0, used = 0;
  void* alloc(size_t n) {
    n = (n + 15) & ~(size_t)15;
    while (chunk < chunks.size() && used + n > sizes[chunk]) { ++chunk; used = 0; }
    if (chunk == chunks.size()) {
      const size_t sz = std::max<size_t>(n, (size_t)64 << 10);
      chunks.emplace_back(new char[sz]);
      sizes.push_back(sz);
      used = 0;
    }
    void* p = chunks[chunk].get() + used;
    used += n;
    return p;
  }
  void reset() {
    if (chunks.size() > 4) { chunks.resize(1); sizes.resize(1); }  // an unusually long input: give the memory back
    chunk = 0;
    used = 0;
  }
};
Arena& arena() { static thread_local Arena a; return a; }
template <class T>
struct ArenaAlloc {
  typedef T value_type;
  ArenaAlloc() = default;
  template <class V> ArenaAlloc(const ArenaAlloc<V>&) {}
  T* allocate(size_t n) { return static_cast<T*>(arena().alloc(n * sizeof(T))); }
  void deallocate(T*, size_t) noexcept {}
  template <class V> bool operator==(const ArenaAlloc<V>&) const { return true; }
  template <class V> bool operator!=(const ArenaAlloc<V>&) const { return false; }
};
// Text inside one edit_script call: a (pointer, length) view of code points that live in the arena or in the two input
// strings.  substr is a view of the same storage, concatenation writes new arena storage; nothing is ever modified in place, so
// views stay valid for the whole call, and a Diff is a trivially copyable 24 bytes (the diff clean-ups insert into and erase
// from the middle of the script all the time).
struct AU {
  static constexpr size_t npos = (size_t)-1;
  const char32_t* p = nullptr;
  size_t n = 0;
  AU() = default;
  AU(const char32_t* q, size_t len) : p(q), n(len) {}
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  const char32_t* data() const { return p; }
  const char32_t* begin() const { return p; }
  const char32_t* end() const { return p + n; }
  char32_t operator[](size_t i) const { return p[i]; }
  char32_t front() const { return p[0]; }
  char32_t back() const { return p[n - 1]; }
  void clear() { n = 0; }
  AU substr(size_t pos, size_t len = npos) const { return AU(p + pos, std::min(len, n - pos)); }
  int compare(size_t pos, size_t len, const AU& o, size_t opos, size_t olen) const {  // equal-length ranges only (common_overlap)
    (void)olen;
    return std::char_traits<char32_t>::compare(p + pos, o.p + opos, len);
  }
  size_t find(const AU& needle) const {
    const char32_t* it = std::search(p, p + n, needle.p, needle.p + needle.n);
    return it == p + n && needle.n ? npos : (size_t)(it - p);
  }
  static AU concat(const char32_t* a, size_t na, const char32_t* b, size_t nb) {
    char32_t* q = static_cast<char32_t*>(arena().alloc((na + nb) * sizeof(char32_t) + 4));
    if (na) memcpy(q, a, na * sizeof(char32_t));
    if (nb) memcpy(q + na, b, nb * sizeof(char32_t));
    return AU(q, na + nb);
  }
  AU& operator+=(const AU& o) { if (o.n) *this = concat(p, n, o.p, o.n); return *this; }
  AU& operator+=(char32_t c) { *this = concat(p, n, &c, 1); return *this; }
};
AU operator+(const AU& a, const AU& b) { return AU::concat(a.p, a.n, b.p, b.n); }
AU operator+(const AU& a, char32_t c) { return AU::concat(a.p, a.n, &c, 1); }
bool operator==(const AU& a, const AU& b) { return a.n == b.n && std::char_traits<char32_t>::compare(a.p, b.p, a.n) == 0; }
bool operator!=(const AU& a, const AU& b) { return !(a == b); }
struct Diff { char op; AU text; };  // '=', '-', '+'
typedef std::vector<Diff, ArenaAlloc<Diff>> Diffs;

U to_u32(const char* s, size_t n) {
  U out;
  out.reserve(n);
  for (size_t i = 0; i < n;) {
    int l;
    out.push_back(utf8_decode_at(s + i, n - i, &l));
    i += (size_t)l;
  }
  return out;
}
U to_u32(const std::string& s) { return to_u32(s.data(), s.size()); }
template <class A, class B>
bool same_text(const A& a, const B& b) { return a.size() == b.size() && std::char_traits<char32_t>::compare(a.data(), b.data(), a.size()) == 0; }
template <class A, class B>
size_t common_prefix(const A& a, const B& b) {
  const size_t n = std::min(a.size(), b.size());
  size_t i = 0;
  while (i < n && a[i] == b[i]) ++i;
  return i;
}
template <class A, class B>
size_t common_suffix(const A& a, const B& b) {
  const size_t n = std::min(a.size(), b.size());
  size_t i = 0;
  while (i < n && a[a.size() - 1 - i] == b[b.size() - 1 - i]) ++i;
  return i;
}
size_t common_overlap(const AU& a, const AU& b) {  // longest suffix of a that is a prefix of b
  for (size_t k = std::min(a.size(), b.size()); k > 0; --k)
    if (a.compare(a.size() - k, k, b, 0, k) == 0) return k;
  return 0;
}
template <class A, class B>
bool ends_with(const A& s, const B& x) {
  return s.size() >= x.size() && std::char_traits<char32_t>::compare(s.data() + (s.size() - x.size()), x.data(), x.size()) == 0;
}
template <class A, class B>
bool starts_with(const A& s, const B& x) { return s.size() >= x.size() && std::char_traits<char32_t>::compare(s.data(), x.data(), x.size()) == 0; }

void cleanup_merge(Diffs& d);
Diffs diff_main(AU a, AU b);

Diffs bisect(const AU& a, const AU& b) {
  const long n1 = (long)a.size(), n2 = (long)b.size();
  const long max_d = (n1 + n2 + 1) / 2, v_offset = max_d, v_length = 2 * max_d;
  std::vector<long, ArenaAlloc<long>> v1((size_t)v_length, -1), v2((size_t)v_length, -1);
  v1[(size_t)v_offset + 1] = 0;
  v2[(size_t)v_offset + 1] = 0;
  const long delta = n1 - n2;
  const bool front = delta % 2 != 0;
  long k1start = 0, k1end = 0, k2start = 0, k2end = 0;
  auto split = [&](long x, long y) {
    Diffs l = diff_main(a.substr(0, (size_t)x), b.substr(0, (size_t)y)), r = diff_main(a.substr((size_t)x), b.substr((size_t)y));
    l.insert(l.end(), r.begin(), r.end());
    return l;
  };
  for (long d = 0; d < max_d; ++d) {
    for (long k1 = -d + k1start; k1 <= d - k1end; k1 += 2) {
      const long k1o = v_offset + k1;
      long x1;
      if (k1 == -d || (k1 != d && v1[(size_t)k1o - 1] < v1[(size_t)k1o + 1])) x1 = v1[(size_t)k1o + 1];
      else x1 = v1[(size_t)k1o - 1] + 1;
      long y1 = x1 - k1;
      while (x1 < n1 && y1 < n2 && a[(size_t)x1] == b[(size_t)y1]) { ++x1; ++y1; }
      v1[(size_t)k1o] = x1;
      if (x1 > n1) k1end += 2;
      else if (y1 > n2) k1start += 2;
      else if (front) {
        const long k2o = v_offset + delta - k1;
        if (k2o >= 0 && k2o < v_length && v2[(size_t)k2o] != -1) {
          const long x2 = n1 - v2[(size_t)k2o];
          if (x1 >= x2) return split(x1, y1);
        }
      }
    }
    for (long k2 = -d + k2start; k2 <= d - k2end; k2 += 2) {
      const long k2o = v_offset + k2;
      long x2;
      if (k2 == -d || (k2 != d && v2[(size_t)k2o - 1] < v2[(size_t)k2o + 1])) x2 = v2[(size_t)k2o + 1];
      else x2 = v2[(size_t)k2o - 1] + 1;
      long y2 = x2 - k2;
      while (x2 < n1 && y2 < n2 && a[(size_t)(n1 - x2 - 1)] == b[(size_t)(n2 - y2 - 1)]) { ++x2; ++y2; }
      v2[(size_t)k2o] = x2;
      if (x2 > n1) k2end += 2;
      else if (y2 > n2) k2start += 2;
      else if (!front) {
        const long k1o = v_offset + delta - k2;
        if (k1o >= 0 && k1o < v_length && v1[(size_t)k1o] != -1) {
          const long x1 = v1[(size_t)k1o], y1 = v_offset + x1 - k1o;
          if (x1 >= n1 - x2) return split(x1, y1);
        }
      }
    }
  }
  return Diffs{{'-', a}, {'+', b}};
}

Diffs compute(const AU& a, const AU& b) {
  if (a.empty()) return b.empty() ? Diffs{} : Diffs{{'+', b}};
  if (b.empty()) return Diffs{{'-', a}};
  const AU& longt = a.size() > b.size() ? a : b;
  const AU& shortt = a.size() > b.size() ? b : a;
  const size_t i = longt.find(shortt);
  if (i != AU::npos) {
    const char op = a.size() > b.size() ? '-' : '+';
    Diffs out;
    if (i) out.push_back({op, longt.substr(0, i)});
    out.push_back({'=', shortt});
    if (i + shortt.size() < longt.size()) out.push_back({op, longt.substr(i + shortt.size())});
    return out;
  }
  if (shortt.size() == 1) return Diffs{{'-', a}, {'+', b}};
  return bisect(a, b);
}

Diffs diff_main(AU a, AU b) {
  if (a == b) return a.empty() ? Diffs{} : Diffs{{'=', a}};
  const size_t p = common_prefix(a, b);
  const AU prefix = a.substr(0, p);
  a = a.substr(p);
  b = b.substr(p);
  const size_t s = common_suffix(a, b);
  const AU suffix = a.substr(a.size() - s);
  a = a.substr(0, a.size() - s);
  b = b.substr(0, b.size() - s);
  Diffs d = compute(a, b);
  if (!prefix.empty()) d.insert(d.begin(), Diff{'=', prefix});
  if (!suffix.empty()) d.push_back(Diff{'=', suffix});
  cleanup_merge(d);
  return d;
}

void cleanup_merge(Diffs& d) {
  d.push_back({'=', AU()});
  size_t pointer = 0;
  size_t count_delete = 0, count_insert = 0;
  AU text_delete, text_insert;
  while (pointer < d.size()) {
    if (d[pointer].op == '+') { ++count_insert; text_insert += d[pointer].text; ++pointer; }
    else if (d[pointer].op == '-') { ++count_delete; text_delete += d[pointer].text; ++pointer; }
    else {
      if (count_delete + count_insert > 1) {
        if (count_delete != 0 && count_insert != 0) {
          size_t cl = common_prefix(text_insert, text_delete);
          if (cl) {
            const long x = (long)pointer - (long)count_delete - (long)count_insert - 1;
            if (x >= 0 && d[(size_t)x].op == '=') d[(size_t)x].text += text_insert.substr(0, cl);
            else { d.insert(d.begin(), Diff{'=', text_insert.substr(0, cl)}); ++pointer; }
            text_insert = text_insert.substr(cl);
            text_delete = text_delete.substr(cl);
          }
          cl = common_suffix(text_insert, text_delete);
          if (cl) {
            d[pointer].text = text_insert.substr(text_insert.size() - cl) + d[pointer].text;
            text_insert = text_insert.substr(0, text_insert.size() - cl);
            text_delete = text_delete.substr(0, text_delete.size() - cl);
          }
        }
        Diffs new_ops;
        if (!text_delete.empty()) new_ops.push_back({'-', text_delete});
        if (!text_insert.empty()) new_ops.push_back({'+', text_insert});
        pointer -= count_delete + count_insert;
        d.erase(d.begin() + (long)pointer, d.begin() + (long)(pointer + count_delete + count_insert));
        d.insert(d.begin() + (long)pointer, new_ops.begin(), new_ops.end());
        pointer += new_ops.size() + 1;
      } else if (pointer != 0 && d[pointer - 1].op == '=') {
        d[pointer - 1].text += d[pointer].text;
        d.erase(d.begin() + (long)pointer);
      } else ++pointer;
      count_insert = count_delete = 0;
      text_delete.clear();
      text_insert.clear();
    }
  }
  if (d.back().text.empty()) d.pop_back();
  bool changes = false;
  pointer = 1;
  while (pointer + 1 < d.size()) {
    if (d[pointer - 1].op == '=' && d[pointer + 1].op == '=') {
      const AU prev_t = d[pointer - 1].text, cur_t = d[pointer].text, next_t = d[pointer + 1].text;
      if (!prev_t.empty() && ends_with(cur_t, prev_t)) {
        d[pointer].text = prev_t + cur_t.substr(0, cur_t.size() - prev_t.size());
        d[pointer + 1].text = prev_t + next_t;
        d.erase(d.begin() + (long)pointer - 1);
        changes = true;
      } else if (!next_t.empty() && starts_with(cur_t, next_t)) {
        d[pointer - 1].text = prev_t + next_t;
        d[pointer].text = cur_t.substr(next_t.size()) + next_t;
        d.erase(d.begin() + (long)pointer + 1);
        changes = true;
      }
    }
    ++pointer;
  }
  if (changes) cleanup_merge(d);
}

bool is_alnum(char32_t c) { return is_alphabetic_cp(c) || (c >= U'0' && c <= U'9'); }  // char::is_alphanumeric, ASCII digits
bool is_space(char32_t c) { return c == U' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000; }
int semantic_score(const AU& one, const AU& two) {
  if (one.empty() || two.empty()) return 6;
  const char32_t c1 = one.back(), c2 = two.front();
  const bool na1 = !is_alnum(c1), na2 = !is_alnum(c2);
  const bool ws1 = na1 && is_space(c1), ws2 = na2 && is_space(c2);
  const bool lb1 = ws1 && (c1 == U'\r' || c1 == U'\n'), lb2 = ws2 && (c2 == U'\r' || c2 == U'\n');
  const bool bl1 = lb1 && (ends_with(one, U(U"\n\n")) || ends_with(one, U(U"\n\r\n")));
  const bool bl2 = lb2 && (starts_with(two, U(U"\n\n")) || starts_with(two, U(U"\r\n\n")) || starts_with(two, U(U"\n\r\n")) || starts_with(two, U(U"\r\n\r\n")));
  if (bl1 || bl2) return 5;
  if (lb1 || lb2) return 4;
  if (na1 && !ws1 && ws2) return 3;
  if (ws1 || ws2) return 2;
  if (na1 || na2) return 1;
  return 0;
}

void cleanup_semantic_lossless(Diffs& d) {
  long pointer = 1;
  while (pointer + 1 < (long)d.size()) {
    if (d[(size_t)pointer - 1].op == '=' && d[(size_t)pointer + 1].op == '=') {
      AU eq1 = d[(size_t)pointer - 1].text, edit = d[(size_t)pointer].text, eq2 = d[(size_t)pointer + 1].text;
      const size_t co = common_suffix(eq1, edit);
      if (co) {
        const AU cs = edit.substr(edit.size() - co);
        eq1 = eq1.substr(0, eq1.size() - co);
        edit = cs + edit.substr(0, edit.size() - co);
        eq2 = cs + eq2;
      }
      AU b1 = eq1, be = edit, b2 = eq2;
      int best = semantic_score(eq1, edit) + semantic_score(edit, eq2);
      while (!edit.empty() && !eq2.empty() && edit[0] == eq2[0]) {
        eq1 += edit[0];
        edit = edit.substr(1) + eq2[0];
        eq2 = eq2.substr(1);
        const int sc = semantic_score(eq1, edit) + semantic_score(edit, eq2);
        if (sc >= best) { best = sc; b1 = eq1; be = edit; b2 = eq2; }
      }
      if (d[(size_t)pointer - 1].text != b1) {
        if (!b1.empty()) d[(size_t)pointer - 1].text = b1;
        else { d.erase(d.begin() + pointer - 1); --pointer; }
        d[(size_t)pointer].text = be;
        if (!b2.empty()) d[(size_t)pointer + 1].text = b2;
        else { d.erase(d.begin() + pointer + 1); --pointer; }
      }
    }
    ++pointer;
  }
}

void cleanup_semantic(Diffs& d) {
  bool changes = false;
  std::vector<long, ArenaAlloc<long>> equalities;
  bool have_last = false;
  AU last_eq;
  long pointer = 0;
  size_t li1 = 0, ld1 = 0, li2 = 0, ld2 = 0;
  while (pointer < (long)d.size()) {
    if (d[(size_t)pointer].op == '=') {
      equalities.push_back(pointer);
      li1 = li2; li2 = 0; ld1 = ld2; ld2 = 0;
      last_eq = d[(size_t)pointer].text;
      have_last = true;
    } else {
      if (d[(size_t)pointer].op == '+') li2 += d[(size_t)pointer].text.size();
      else ld2 += d[(size_t)pointer].text.size();
      if (have_last && !last_eq.empty() && last_eq.size() <= std::max(li1, ld1) && last_eq.size() <= std::max(li2, ld2)) {
        const long e = equalities.back();
        d.insert(d.begin() + e, Diff{'-', last_eq});
        d[(size_t)e + 1].op = '+';
        equalities.pop_back();
        if (!equalities.empty()) equalities.pop_back();
        pointer = equalities.empty() ? -1 : equalities.back();
        li1 = ld1 = li2 = ld2 = 0;
        have_last = false;
        last_eq.clear();
        changes = true;
      }
    }
    ++pointer;
  }
  if (changes) cleanup_merge(d);
  cleanup_semantic_lossless(d);
  size_t p = 1;
  while (p < d.size()) {
    if (d[p - 1].op == '-' && d[p].op == '+') {
      const AU deletion = d[p - 1].text, insertion = d[p].text;
      const size_t o1 = common_overlap(deletion, insertion), o2 = common_overlap(insertion, deletion);
      if (o1 >= o2) {
        if (2 * o1 >= deletion.size() || 2 * o1 >= insertion.size()) {
          d.insert(d.begin() + (long)p, Diff{'=', insertion.substr(0, o1)});
          d[p - 1].text = deletion.substr(0, deletion.size() - o1);
          d[p + 1].text = insertion.substr(o1);
          ++p;
        }
      } else if (2 * o2 >= deletion.size() || 2 * o2 >= insertion.size()) {
        d.insert(d.begin() + (long)p, Diff{'=', deletion.substr(0, o2)});
        d[p - 1] = Diff{'+', insertion.substr(0, insertion.size() - o2)};
        d[p + 1] = Diff{'-', deletion.substr(o2)};
        ++p;
      }
      ++p;
    }
    ++p;
  }
}

Diffs edit_script(const U& a, const U& b) {  // the result lives in the calling thread's arena until its next call
  arena().reset();
  Diffs d = diff_main(AU(a.data(), a.size()), AU(b.data(), b.size()));  // views of the inputs: they outlive the call
  cleanup_semantic(d);
  cleanup_merge(d);
  d.erase(std::remove_if(d.begin(), d.end(), [](const Diff& x) { return x.text.empty(); }), d.end());
  return d;
}

template <class S>
std::string to_utf8(const S& s) {
  std::string out;
  for (char32_t c : s) {
    if (c < 0x80) out.push_back((char)c);
    else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) { out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else { out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
  }
  return out;
}

bool found_in(const Confusable& c, const Diffs& ref) {  // src/confusables.rs:47-127
  const size_t l = c.ops.size();
  size_t matches = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    if (matches >= l) continue;
    const char op = c.ops[matches];
    bool found = false;
    if (op == ref[i].op)
      for (const U& s : c.options[matches]) {
        bool ok;
        if (op != '=') ok = ends_with(ref[i].text, s);
        else if (matches == 0 && matches == l - 1) ok = same_text(s, ref[i].text);
        else if (matches == 0) ok = ends_with(ref[i].text, s);
        else if (matches == l - 1) ok = starts_with(ref[i].text, s);
        else ok = same_text(s, ref[i].text);
        if (ok) { found = true; break; }
      }
    if (!found) {
      matches = 0;
      if (c.strictbegin) return false;
      continue;
    }
    if (++matches == l) return c.strictend ? i == ref.size() - 1 : true;
  }
  return false;
}

}  // namespace

std::string edit_script_string(const std::string& source, const std::string& target) {
  std::string out;
  const U a = to_u32(source), b = to_u32(target);  // the script holds views of them
  for (const Diff& x : edit_script(a, b)) {
    out.push_back(x.op);
    out.push_back('[');
    out += to_utf8(x.text);
    out.push_back(']');
  }
  return out;
}

int HostModel::add_to_confusables(const std::string& script, double weight, std::string& err) {  // src/lib.rs:446-458
  if (script.empty()) { err = "empty confusable pattern"; return ANX_EINVAL; }
  Confusable c;
  c.weight = weight;
  c.strictbegin = script.front() == '^';
  c.strictend = script.back() == '$';
  const std::string body = script.substr(c.strictbegin ? 1 : 0, script.size() - (c.strictbegin ? 1 : 0) - (c.strictend ? 1 : 0));
  size_t begin = 0;
  for (size_t i = 0; i < body.size(); ++i)
    if (body[i] == ']') {
      const std::string ins = body.substr(begin, i + 1 - begin);
      if (ins.size() <= 3 || ins[1] != '[' || (ins[0] != '=' && ins[0] != '+' && ins[0] != '-')) {
        err = "invalid edit instruction '" + ins + "'";
        return ANX_EINVAL;
      }
      c.ops.push_back(ins[0]);
      std::vector<U> opts;
      const std::string inner = ins.substr(2, ins.size() - 3);
      size_t pos = 0;
      for (;;) {
        const size_t e = inner.find('|', pos);
        opts.push_back(to_u32(inner.substr(pos, e == std::string::npos ? std::string::npos : e - pos)));
        if (e == std::string::npos) break;
        pos = e + 1;
      }
      Confusable::Screen sc;
      sc.simple = true;
      for (const U& o : opts) {
        if (o.size() == 1 && o[0] < 128) sc.bits[o[0] >> 6] |= 1ull << (o[0] & 63);
        else sc.simple = false;
      }
      c.screen.push_back(sc);
      c.options.push_back(std::move(opts));
      begin = i + 1;
    }
  if (c.ops.empty()) { err = "confusable pattern without instructions"; return ANX_EINVAL; }
  confusables.push_back(std::move(c));
  return ANX_OK;
}

int HostModel::read_confusablelist(const std::string& path, std::string& err) {  // src/lib.rs:409-443
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open " + path; return ANX_EIO; }
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    const size_t tab = line.find('\t');
    double w = 1.0;
    if (tab != std::string::npos) {
      const size_t tab2 = line.find('\t', tab + 1);
      const std::string field = line.substr(tab + 1, tab2 == std::string::npos ? std::string::npos : tab2 - tab - 1);
      char* endp = nullptr;
      w = strtod(field.c_str(), &endp);
      if (endp == field.c_str() || *endp) { err = "score should be a float: '" + field + "'"; return ANX_EINVAL; }
    }
    const int rc = add_to_confusables(line.substr(0, tab), w, err);
    if (rc) return rc;
  }
  return ANX_OK;
}

namespace {
// A pattern can only be found in the edit script of (input -> candidate) if every instruction has an option that occurs in the
// string the instruction's text is taken from: a deletion's text is a piece of the input, an insertion's a piece of the
// candidate, an equality's a piece of both (found_in compares the option with the end / start / whole of that text); with `$`
// the last diff is the end of the strings.  Necessary, not sufficient: it only decides whether the edit script is worth
// computing -- on BASELINE configs[2] most ranked rows cannot match any of the patterns.
struct CharSet { uint64_t w[2] = {0, 0}; bool other = false; };  // ASCII presence bits; `other`: some non-ASCII character
CharSet charset_of(const U& s) {
  CharSet cs;
  for (char32_t ch : s) {
    if (ch < 128) cs.w[ch >> 6] |= 1ull << (ch & 63);
    else cs.other = true;
  }
  return cs;
}
bool occurs(const U& opt, const U& s, const CharSet& cs) {
  if (opt.size() == 1 && opt[0] < 128) return (cs.w[opt[0] >> 6] >> (opt[0] & 63)) & 1ull;
  return s.find(opt) != U::npos;
}
bool may_match(const Confusable& c, const U& in, const CharSet& ins, const U& cand, const CharSet& cs) {
  const size_t l = c.ops.size();
  for (size_t k = 0; k < l; ++k) {
    const bool tail = c.strictend && k == l - 1 && c.ops[k] != '=';
    if (c.screen[k].simple && !tail) {  // one-character options: the same test on the presence bits of the two strings
      const uint64_t* b = c.screen[k].bits;
      uint64_t h0 = b[0], h1 = b[1];
      if (c.ops[k] != '+') { h0 &= ins.w[0]; h1 &= ins.w[1]; }
      if (c.ops[k] != '-') { h0 &= cs.w[0]; h1 &= cs.w[1]; }
      if (!(h0 | h1)) return false;
      continue;
    }
    bool any = false;
    for (const U& opt : c.options[k]) {
      bool ok = (c.ops[k] == '+' || occurs(opt, in, ins)) && (c.ops[k] == '-' || occurs(opt, cand, cs));
      if (ok && c.strictend && k == l - 1 && c.ops[k] != '=')  // the last diff ends the script: its text ends the string
        ok = ends_with(c.ops[k] == '+' ? cand : in, opt);
      if (ok) { any = true; break; }
    }
    if (!any) return false;
  }
  return true;
}
}  // namespace

struct HostModel::ConfCache {
  std::vector<U> text;
  std::vector<CharSet> cs;
};

void HostModel::confusable_weights(const char* input, size_t len, const uint64_t* ids, size_t n, double* out) const {  // src/lib.rs:1733-1756
  const ConfCache* held = conf_cache.load(std::memory_order_acquire);
  if (!held || held->text.size() != decoder.size()) {  // first use, or the vocabulary grew since (items are only ever appended)
    std::lock_guard<std::mutex> g(conf_cache_mu);
    held = conf_cache.load(std::memory_order_acquire);
    if (!held || held->text.size() != decoder.size()) {
      auto cc = std::make_shared<ConfCache>();
      cc->text.resize(decoder.size());
      cc->cs.resize(decoder.size());
      for (size_t i = 0; i < decoder.size(); ++i) { cc->text[i] = to_u32(decoder[i].text); cc->cs[i] = charset_of(cc->text[i]); }
      conf_cache_owned.push_back(cc);
      conf_cache.store(cc.get(), std::memory_order_release);
      held = cc.get();
    }
  }
  const ConfCache& cc = *held;
  const U in = to_u32(input, len);
  const CharSet ins = charset_of(in);
  // the patterns whose deletions / equalities the INPUT can supply at all: the rows of an input are only screened against these
  const Confusable* live[64];
  size_t nlive = 0;
  bool all_live = confusables.size() > 64;
  if (!all_live)
    for (const Confusable& c : confusables) {
      bool ok = true;
      for (size_t j = 0; j < c.ops.size() && ok; ++j)
        if (c.ops[j] != '+' && c.screen[j].simple) ok = ((c.screen[j].bits[0] & ins.w[0]) | (c.screen[j].bits[1] & ins.w[1])) != 0;
      if (ok) live[nlive++] = &c;
    }
  for (size_t k = 0; k < n; ++k) {
    double weight = 1.0;
    if (ids[k] < cc.text.size()) {
      const U& cand = cc.text[ids[k]];
      bool any = false;
      if (all_live) { for (const Confusable& c : confusables) any = any || may_match(c, in, ins, cand, cc.cs[ids[k]]); }
      else for (size_t j = 0; j < nlive && !any; ++j) any = may_match(*live[j], in, ins, cand, cc.cs[ids[k]]);
      if (any) {  // else no pattern can be in the script: not computed
        const Diffs script = edit_script(in, cand);
        for (const Confusable& c : confusables)
          if (found_in(c, script)) weight *= c.weight;
      }
    }
    out[k] = weight;
  }
}

double HostModel::confusable_weight(const std::string& input, uint64_t candidate) const {
  double w = 1.0;
  confusable_weights(input, &candidate, 1, &w);
  return w;
}

}  // namespace anx
