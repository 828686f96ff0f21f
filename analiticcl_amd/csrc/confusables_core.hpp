// confusables_core.hpp -- the edit script behind confusable weighting and the pattern matcher, as ONE body of code that is
// compiled for the host (confusables.cpp) and for the device (conf.hip: one lane per ranked row).
//
// The reference obtains the script from sesdiff::shortest_edit_script(input, candidate, false, false, false)
// (src/lib.rs:1736; sesdiff 0.3.1), which maps dissimilar::diff chunks 1:1 to Identity / Deletion / Insertion.  Neither crate is
// in the reference tree: this restates the published algorithm (dissimilar = the Diff part of Google's diff-match-patch: common
// prefix / suffix, containment speed-up, Myers bisect, cleanup_semantic incl. lossless shifts and overlap extraction,
// cleanup_merge; on Unicode scalar values).  PARITY UNPINNED beyond tests/main.rs:914-1020.  Pattern matching:
// src/confusables.rs:47-127 (found_in).
//
// No allocation, no recursion, no library calls.  Everything a call works on -- the two inputs, a bump arena, the diffs, the
// diagonal arrays of the bisect, the frame stack that replaces the recursion of diff_main -- lives in ONE array of 32-bit words, the
// context's "lane memory", and is addressed by word index: a text is an (index, length) view, a diff three words, a frame eight.
// Word w of the memory is mem[w * S]: S = 1 on the host; on the device the 64 lanes of a wave interleave their memories (S = 64), so
// that lanes working on the same word of their own rows -- the common case: every row starts out at the same offsets -- share cache
// lines instead of touching one line each (round 3: k_conf_script was bound by exactly that, 64 lines per memory instruction).
// The diffs of nested diff_main calls live back to back (the result of the left half of a bisect split, then the right half:
// concatenation is free).  A context that runs out of room sets `overflow` and the result is void: the host retries with larger
// buffers (so it never fails), the device hands the row to the host.  Pattern options live in a shared pool outside the lane
// memory (PView: a plain pointer).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define ANX_HD __host__ __device__ inline
#define ANX_HDS __host__ __device__ static inline
#else
#define ANX_HD inline
#define ANX_HDS static inline
#endif

namespace anx {
namespace cdiff {

typedef uint32_t cp_t;  // Unicode scalar value

struct View {   // a run of code points in the lane memory: word index + length; never modified through a view
  uint32_t p, n;
};
struct PView {  // a run of code points in shared memory (pattern options)
  const cp_t* p;
  uint32_t n;
};
struct Diff {   // as the callers see one: three words of lane memory {op, text.p, text.n}
  uint32_t op;  // '=', '-', '+'
  View text;
};
struct Frame {  // one diff_main call: eight words of lane memory
  View a, b;            // phase 0: the call's arguments; phase 1: the right halves of the bisect split
  View suffix;
  uint32_t seg;         // first diff of this call's result
  uint32_t phase;
};
constexpr uint32_t DIFF_WORDS = 3, FRAME_WORDS = 8;
struct Ctx {
  uint32_t* mem;        // word w of the lane memory = mem[w * S] (Core<S>)
  uint32_t arena_off, arena_cap, arena_used;  // bump arena of code points: words [arena_off, arena_off + arena_cap)
  uint32_t d_off, d_cap, nd;                  // diff stack: diff i = words d_off + 3 i ..
  uint32_t v_off, v_cap;                      // bisect: the two diagonal arrays (2 x v_length); cleanup_semantic: the equalities stack
  uint32_t f_off, frame_cap;                  // frame i = words f_off + 8 i ..
  const uint32_t (*alpha)[2];  // inclusive code point ranges of char::is_alphabetic, sorted
  uint32_t nalpha;
  bool overflow;
};

// ---- patterns (flattened: the same tables on host and device) -----------------------------------------------------------------
struct FlatOp {         // one edit instruction of a pattern
  uint32_t op;          // '=', '+', '-'
  uint32_t simple;      // every option is one ASCII character: `bits` are their presence bits
  uint64_t bits[2];
  uint32_t opt_begin, nopt;  // options: entries of the option table
};
struct FlatOpt { uint32_t off, len; };  // code points in the option pool
struct FlatConf {       // src/confusables.rs:5-11
  double weight;
  uint32_t op_begin, nops;
  uint32_t strictbegin, strictend;
};
struct Patterns {
  const FlatConf* conf;
  uint32_t nconf;
  const FlatOp* ops;
  const FlatOpt* opts;
  const cp_t* pool;
};
struct CharSet {  // ASCII presence bits of a string; `other`: it has a non-ASCII character
  uint64_t w[2];
  uint32_t other;
};

ANX_HD View mk(uint32_t p, uint32_t n) { View v; v.p = p; v.n = n; return v; }
ANX_HD PView pmk(const cp_t* p, uint32_t n) { PView v; v.p = p; v.n = n; return v; }
ANX_HD View sub(const View& s, uint32_t pos) { return mk(s.p + pos, s.n - pos); }
ANX_HD View sub(const View& s, uint32_t pos, uint32_t len) { return mk(s.p + pos, len < s.n - pos ? len : s.n - pos); }
ANX_HD bool is_space(cp_t ch) {
  return ch == ' ' || (ch >= 9 && ch <= 13) || ch == 0x85 || ch == 0xA0 || ch == 0x1680 || (ch >= 0x2000 && ch <= 0x200A) || ch == 0x2028 ||
         ch == 0x2029 || ch == 0x202F || ch == 0x205F || ch == 0x3000;
}
ANX_HD bool is_alphabetic(const Ctx& c, cp_t cp) {
  if (cp < 128) return ((cp | 32u) - 'a') < 26u;  // the ASCII part of the table: the letters
  int32_t lo = 0, hi = (int32_t)c.nalpha - 1;
  while (lo <= hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (cp < c.alpha[mid][0]) hi = mid - 1;
    else if (cp > c.alpha[mid][1]) lo = mid + 1;
    else return true;
  }
  return false;
}
ANX_HD bool is_alnum(const Ctx& c, cp_t ch) { return is_alphabetic(c, ch) || (ch >= '0' && ch <= '9'); }  // char::is_alphanumeric, ASCII digits

ANX_HD CharSet charset_of_array(const cp_t* s, uint32_t n) {  // of a string outside any lane memory (the vocabulary cache)
  CharSet cs;
  cs.w[0] = cs.w[1] = 0;
  cs.other = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (s[i] < 64) cs.w[0] |= 1ull << s[i];  // (no run-time index into the two words: on the device that became a hidden 3 KB LDS array)
    else if (s[i] < 128) cs.w[1] |= 1ull << (s[i] & 63);
    else cs.other = 1;
  }
  return cs;
}

template <uint32_t S, uint32_t G = 1>
struct Core {
// ---- the lane memory: S lanes interleave in groups of G words ----------------------------------------------------------------
ANX_HDS uint32_t& W(const Ctx& c, uint32_t w) { return c.mem[(size_t)(w / G) * (S * G) + (w % G)]; }
ANX_HDS cp_t at(const Ctx& c, const View& s, uint32_t i) { return W(c, s.p + i); }
ANX_HDS uint32_t d_op(const Ctx& c, uint32_t i) { return W(c, c.d_off + DIFF_WORDS * i); }
ANX_HDS View d_text(const Ctx& c, uint32_t i) { return mk(W(c, c.d_off + DIFF_WORDS * i + 1), W(c, c.d_off + DIFF_WORDS * i + 2)); }
ANX_HDS uint32_t d_len(const Ctx& c, uint32_t i) { return W(c, c.d_off + DIFF_WORDS * i + 2); }
ANX_HDS void set_op(const Ctx& c, uint32_t i, uint32_t op) { W(c, c.d_off + DIFF_WORDS * i) = op; }
ANX_HDS void set_text(const Ctx& c, uint32_t i, const View& t) { W(c, c.d_off + DIFF_WORDS * i + 1) = t.p; W(c, c.d_off + DIFF_WORDS * i + 2) = t.n; }
ANX_HDS void d_move(const Ctx& c, uint32_t to, uint32_t from) {
  for (uint32_t k = 0; k < DIFF_WORDS; ++k) W(c, c.d_off + DIFF_WORDS * to + k) = W(c, c.d_off + DIFF_WORDS * from + k);
}
ANX_HDS Diff diff_at(const Ctx& c, uint32_t i) { Diff d; d.op = d_op(c, i); d.text = d_text(c, i); return d; }
ANX_HDS View empty(const Ctx& c) { return mk(c.arena_off, 0); }
ANX_HDS Frame load_frame(const Ctx& c, uint32_t i) {
  const uint32_t o = c.f_off + FRAME_WORDS * i;
  Frame f;
  f.a = mk(W(c, o), W(c, o + 1)); f.b = mk(W(c, o + 2), W(c, o + 3)); f.suffix = mk(W(c, o + 4), W(c, o + 5));
  f.seg = W(c, o + 6); f.phase = W(c, o + 7);
  return f;
}
ANX_HDS void store_frame(const Ctx& c, uint32_t i, const Frame& f) {
  const uint32_t o = c.f_off + FRAME_WORDS * i;
  W(c, o) = f.a.p; W(c, o + 1) = f.a.n; W(c, o + 2) = f.b.p; W(c, o + 3) = f.b.n; W(c, o + 4) = f.suffix.p; W(c, o + 5) = f.suffix.n;
  W(c, o + 6) = f.seg; W(c, o + 7) = f.phase;
}

// ---- texts ------------------------------------------------------------------------------------------------------------
ANX_HDS bool eq_range(const Ctx& c, uint32_t a, uint32_t b, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (W(c, a + i) != W(c, b + i)) return false;
  return true;
}
ANX_HDS bool eq_range(const Ctx& c, uint32_t a, const cp_t* b, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (W(c, a + i) != b[i]) return false;
  return true;
}
ANX_HDS bool same(const Ctx& c, const View& a, const View& b) { return a.n == b.n && eq_range(c, a.p, b.p, a.n); }
ANX_HDS bool same(const Ctx& c, const PView& a, const View& b) { return a.n == b.n && eq_range(c, b.p, a.p, a.n); }
ANX_HDS uint32_t common_prefix(const Ctx& c, const View& a, const View& b) {
  const uint32_t n = a.n < b.n ? a.n : b.n;
  uint32_t i = 0;
  while (i < n && at(c, a, i) == at(c, b, i)) ++i;
  return i;
}
ANX_HDS uint32_t common_suffix(const Ctx& c, const View& a, const View& b) {
  const uint32_t n = a.n < b.n ? a.n : b.n;
  uint32_t i = 0;
  while (i < n && at(c, a, a.n - 1 - i) == at(c, b, b.n - 1 - i)) ++i;
  return i;
}
ANX_HDS uint32_t common_overlap(const Ctx& c, const View& a, const View& b) {  // longest suffix of a that is a prefix of b
  for (uint32_t k = a.n < b.n ? a.n : b.n; k > 0; --k)
    if (eq_range(c, a.p + (a.n - k), b.p, k)) return k;
  return 0;
}
ANX_HDS bool ends_with(const Ctx& c, const View& s, const View& x) { return s.n >= x.n && eq_range(c, s.p + (s.n - x.n), x.p, x.n); }
ANX_HDS bool starts_with(const Ctx& c, const View& s, const View& x) { return s.n >= x.n && eq_range(c, s.p, x.p, x.n); }
ANX_HDS bool ends_with(const Ctx& c, const View& s, const PView& x) { return s.n >= x.n && eq_range(c, s.p + (s.n - x.n), x.p, x.n); }
ANX_HDS bool starts_with(const Ctx& c, const View& s, const PView& x) { return s.n >= x.n && eq_range(c, s.p, x.p, x.n); }
ANX_HDS int64_t find(const Ctx& c, const View& hay, const View& needle) {  // first occurrence, -1 = none (an empty needle is found at 0)
  if (needle.n > hay.n) return -1;
  for (uint32_t i = 0; i + needle.n <= hay.n; ++i)
    if (eq_range(c, hay.p + i, needle.p, needle.n)) return (int64_t)i;
  return -1;
}
ANX_HDS int64_t find(const Ctx& c, const View& hay, const PView& needle) {
  if (needle.n > hay.n) return -1;
  for (uint32_t i = 0; i + needle.n <= hay.n; ++i)
    if (eq_range(c, hay.p + i, needle.p, needle.n)) return (int64_t)i;
  return -1;
}

// ---- arena and diff array ----------------------------------------------------------------------------------------------
ANX_HDS View concat(Ctx& c, const View& a, const View& b) {
  // no copy when there is nothing to join, or when the two pieces already sit side by side (pieces of one string that the
  // clean-ups put back together: the common case)
  if (a.n == 0) return b;
  if (b.n == 0) return a;
  if (a.p + a.n == b.p) return mk(a.p, a.n + b.n);
  const uint32_t n = a.n + b.n;
  if (c.overflow || c.arena_used + n > c.arena_cap) { c.overflow = true; return empty(c); }
  const uint32_t q = c.arena_off + c.arena_used;
  c.arena_used += n;
  for (uint32_t i = 0; i < a.n; ++i) W(c, q + i) = at(c, a, i);
  for (uint32_t i = 0; i < b.n; ++i) W(c, q + a.n + i) = at(c, b, i);
  return mk(q, n);
}
ANX_HDS View append(Ctx& c, const View& a, const View& b) { return b.n ? concat(c, a, b) : a; }  // AU::operator+=
ANX_HDS void d_push(Ctx& c, uint32_t op, const View& t) {
  if (c.overflow || c.nd >= c.d_cap) { c.overflow = true; return; }
  set_op(c, c.nd, op);
  set_text(c, c.nd, t);
  ++c.nd;
}
ANX_HDS void d_insert(Ctx& c, uint32_t pos, uint32_t op, const View& t) {  // before index pos (absolute)
  if (c.overflow || c.nd >= c.d_cap) { c.overflow = true; return; }
  for (uint32_t i = c.nd; i > pos; --i) d_move(c, i, i - 1);
  set_op(c, pos, op);
  set_text(c, pos, t);
  ++c.nd;
}
ANX_HDS void d_erase(Ctx& c, uint32_t pos, uint32_t count) {
  if (c.overflow) return;
  for (uint32_t i = pos; i + count < c.nd; ++i) d_move(c, i, i + count);
  c.nd -= count;
}

// ---- cleanup_merge over the diffs [seg, nd) ----------------------------------------------------------------------------
ANX_HDS void cleanup_merge(Ctx& c, uint32_t seg) {
  for (;;) {
    if (c.overflow) return;
    d_push(c, '=', empty(c));
    if (c.overflow) return;
    uint32_t pointer = seg;
    uint32_t count_delete = 0, count_insert = 0;
    View text_delete = empty(c), text_insert = empty(c);
    while (pointer < c.nd) {
      if (c.overflow) return;
      const uint32_t op = d_op(c, pointer);
      if (op == '+') { ++count_insert; text_insert = append(c, text_insert, d_text(c, pointer)); ++pointer; }
      else if (op == '-') { ++count_delete; text_delete = append(c, text_delete, d_text(c, pointer)); ++pointer; }
      else {
        if (count_delete + count_insert > 1) {
          if (count_delete != 0 && count_insert != 0) {
            uint32_t cl = common_prefix(c, text_insert, text_delete);
            if (cl) {
              const int64_t x = (int64_t)pointer - (int64_t)count_delete - (int64_t)count_insert - 1;
              if (x >= (int64_t)seg && d_op(c, (uint32_t)x) == '=') set_text(c, (uint32_t)x, append(c, d_text(c, (uint32_t)x), sub(text_insert, 0, cl)));
              else { d_insert(c, seg, '=', sub(text_insert, 0, cl)); ++pointer; }
              text_insert = sub(text_insert, cl);
              text_delete = sub(text_delete, cl);
            }
            cl = common_suffix(c, text_insert, text_delete);
            if (cl && !c.overflow) {
              set_text(c, pointer, concat(c, sub(text_insert, text_insert.n - cl), d_text(c, pointer)));
              text_insert = sub(text_insert, 0, text_insert.n - cl);
              text_delete = sub(text_delete, 0, text_delete.n - cl);
            }
          }
          if (c.overflow) return;
          pointer -= count_delete + count_insert;
          d_erase(c, pointer, count_delete + count_insert);
          uint32_t added = 0;
          if (text_delete.n) { d_insert(c, pointer + added, '-', text_delete); ++added; }
          if (text_insert.n) { d_insert(c, pointer + added, '+', text_insert); ++added; }
          pointer += added + 1;
        } else if (pointer != seg && d_op(c, pointer - 1) == '=') {
          set_text(c, pointer - 1, append(c, d_text(c, pointer - 1), d_text(c, pointer)));
          d_erase(c, pointer, 1);
        } else ++pointer;
        count_insert = count_delete = 0;
        text_delete = empty(c);
        text_insert = empty(c);
      }
    }
    if (c.overflow) return;
    if (c.nd > seg && d_len(c, c.nd - 1) == 0) --c.nd;
    bool changes = false;
    pointer = seg + 1;
    while (pointer + 1 < c.nd) {
      if (c.overflow) return;
      if (d_op(c, pointer - 1) == '=' && d_op(c, pointer + 1) == '=') {
        const View prev_t = d_text(c, pointer - 1), cur_t = d_text(c, pointer), next_t = d_text(c, pointer + 1);
        if (prev_t.n && ends_with(c, cur_t, prev_t)) {
          set_text(c, pointer, concat(c, prev_t, sub(cur_t, 0, cur_t.n - prev_t.n)));
          set_text(c, pointer + 1, concat(c, prev_t, next_t));
          d_erase(c, pointer - 1, 1);
          changes = true;
        } else if (next_t.n && starts_with(c, cur_t, next_t)) {
          set_text(c, pointer - 1, concat(c, prev_t, next_t));
          set_text(c, pointer, concat(c, sub(cur_t, next_t.n), next_t));
          d_erase(c, pointer + 1, 1);
          changes = true;
        }
      }
      ++pointer;
    }
    if (!changes) return;
  }
}

// ---- Myers bisect: the split point of the middle snake, or none ---------------------------------------------------------
ANX_HDS bool bisect_split(Ctx& c, const View& a, const View& b, uint32_t* sx, uint32_t* sy) {
  const int32_t n1 = (int32_t)a.n, n2 = (int32_t)b.n;
  const int32_t max_d = (n1 + n2 + 1) / 2, v_offset = max_d, v_length = 2 * max_d;
  if ((uint32_t)(2 * v_length) > c.v_cap) { c.overflow = true; return false; }
  const uint32_t v1 = c.v_off, v2 = c.v_off + (uint32_t)v_length;  // the diagonal arrays: words of lane memory, read as int32
  auto V = [&](uint32_t base, int32_t i) -> int32_t& { return reinterpret_cast<int32_t&>(W(c, base + (uint32_t)i)); };
  for (int32_t i = 0; i < v_length; ++i) { V(v1, i) = -1; V(v2, i) = -1; }
  // (max_d >= 1 here: both strings are non-empty; v_offset + 1 < v_length needs max_d >= 2, which holds for n1 + n2 >= 3;
  // two one-character strings never reach bisect: compute() handles a one-character shorter string)
  if (v_offset + 1 < v_length) { V(v1, v_offset + 1) = 0; V(v2, v_offset + 1) = 0; }
  const int32_t delta = n1 - n2;
  const bool front = delta % 2 != 0;
  int32_t k1start = 0, k1end = 0, k2start = 0, k2end = 0;
  for (int32_t d = 0; d < max_d; ++d) {
    for (int32_t k1 = -d + k1start; k1 <= d - k1end; k1 += 2) {
      const int32_t k1o = v_offset + k1;
      int32_t x1;
      if (k1 == -d || (k1 != d && V(v1, k1o - 1) < V(v1, k1o + 1))) x1 = V(v1, k1o + 1);
      else x1 = V(v1, k1o - 1) + 1;
      int32_t y1 = x1 - k1;
      while (x1 < n1 && y1 < n2 && at(c, a, (uint32_t)x1) == at(c, b, (uint32_t)y1)) { ++x1; ++y1; }
      V(v1, k1o) = x1;
      if (x1 > n1) k1end += 2;
      else if (y1 > n2) k1start += 2;
      else if (front) {
        const int32_t k2o = v_offset + delta - k1;
        if (k2o >= 0 && k2o < v_length && V(v2, k2o) != -1) {
          const int32_t x2 = n1 - V(v2, k2o);
          if (x1 >= x2) { *sx = (uint32_t)x1; *sy = (uint32_t)y1; return true; }
        }
      }
    }
    for (int32_t k2 = -d + k2start; k2 <= d - k2end; k2 += 2) {
      const int32_t k2o = v_offset + k2;
      int32_t x2;
      if (k2 == -d || (k2 != d && V(v2, k2o - 1) < V(v2, k2o + 1))) x2 = V(v2, k2o + 1);
      else x2 = V(v2, k2o - 1) + 1;
      int32_t y2 = x2 - k2;
      while (x2 < n1 && y2 < n2 && at(c, a, (uint32_t)(n1 - x2 - 1)) == at(c, b, (uint32_t)(n2 - y2 - 1))) { ++x2; ++y2; }
      V(v2, k2o) = x2;
      if (x2 > n1) k2end += 2;
      else if (y2 > n2) k2start += 2;
      else if (!front) {
        const int32_t k1o = v_offset + delta - k2;
        if (k1o >= 0 && k1o < v_length && V(v1, k1o) != -1) {
          const int32_t x1 = V(v1, k1o), y1 = v_offset + x1 - k1o;
          if (x1 >= n1 - x2) { *sx = (uint32_t)x1; *sy = (uint32_t)y1; return true; }
        }
      }
    }
  }
  return false;
}

// ---- diff_main: the recursion over bisect splits as a frame stack; result = diffs [seg0, nd) ------------------------------
ANX_HDS void diff_main(Ctx& c, const View& a0, const View& b0) {
  uint32_t nf = 0;
  if (c.frame_cap == 0) { c.overflow = true; return; }
  {
    Frame f0;
    f0.a = a0; f0.b = b0; f0.suffix = empty(c); f0.seg = 0; f0.phase = 0;
    store_frame(c, 0, f0);
  }
  nf = 1;
  while (nf && !c.overflow) {
    Frame f = load_frame(c, nf - 1);
    if (f.phase == 0) {
      View a = f.a, b = f.b;
      f.seg = c.nd;
      if (same(c, a, b)) {  // (returns before the clean-up, like the original)
        if (a.n) d_push(c, '=', a);
        --nf;
        continue;
      }
      const uint32_t p = common_prefix(c, a, b);
      const View prefix = sub(a, 0, p);
      a = sub(a, p);
      b = sub(b, p);
      const uint32_t s = common_suffix(c, a, b);
      f.suffix = sub(a, a.n - s);
      a = sub(a, 0, a.n - s);
      b = sub(b, 0, b.n - s);
      if (prefix.n) d_push(c, '=', prefix);
      f.phase = 2;  // unless a split sends us through the two halves first
      store_frame(c, nf - 1, f);
      // compute()
      if (a.n == 0) { if (b.n) d_push(c, '+', b); continue; }
      if (b.n == 0) { d_push(c, '-', a); continue; }
      const bool a_longer = a.n > b.n;
      const View longt = a_longer ? a : b, shortt = a_longer ? b : a;
      const int64_t i = find(c, longt, shortt);
      if (i >= 0) {
        const uint32_t op = a_longer ? '-' : '+';
        if (i) d_push(c, op, sub(longt, 0, (uint32_t)i));
        d_push(c, '=', shortt);
        if ((uint32_t)i + shortt.n < longt.n) d_push(c, op, sub(longt, (uint32_t)i + shortt.n));
        continue;
      }
      if (shortt.n == 1) { d_push(c, '-', a); d_push(c, '+', b); continue; }
      uint32_t x = 0, y = 0;
      if (!bisect_split(c, a, b, &x, &y)) { d_push(c, '-', a); d_push(c, '+', b); continue; }
      if (nf >= c.frame_cap) { c.overflow = true; return; }
      f.phase = 1;
      f.a = sub(a, x);  // the right halves wait in this frame
      f.b = sub(b, y);
      store_frame(c, nf - 1, f);
      Frame l;
      l.a = sub(a, 0, x);
      l.b = sub(b, 0, y);
      l.suffix = empty(c); l.seg = 0;
      l.phase = 0;
      store_frame(c, nf++, l);
    } else if (f.phase == 1) {  // left half done (its diffs are on the stack): the right half
      if (nf >= c.frame_cap) { c.overflow = true; return; }
      f.phase = 2;
      store_frame(c, nf - 1, f);
      Frame r;
      r.a = f.a;
      r.b = f.b;
      r.suffix = empty(c); r.seg = 0;
      r.phase = 0;
      store_frame(c, nf++, r);
    } else {
      if (f.suffix.n) d_push(c, '=', f.suffix);
      cleanup_merge(c, f.seg);
      --nf;
    }
  }
}

// ---- cleanup_semantic ---------------------------------------------------------------------------------------------------
ANX_HDS bool tail_is(const Ctx& c, const View& s, const cp_t* x, uint32_t n) { return s.n >= n && eq_range(c, s.p + (s.n - n), x, n); }
ANX_HDS bool head_is(const Ctx& c, const View& s, const cp_t* x, uint32_t n) { return s.n >= n && eq_range(c, s.p, x, n); }
ANX_HDS int semantic_score(const Ctx& c, const View& one, const View& two) {
  if (one.n == 0 || two.n == 0) return 6;
  const cp_t c1 = at(c, one, one.n - 1), c2 = at(c, two, 0);
  const bool na1 = !is_alnum(c, c1), na2 = !is_alnum(c, c2);
  const bool ws1 = na1 && is_space(c1), ws2 = na2 && is_space(c2);
  const bool lb1 = ws1 && (c1 == '\r' || c1 == '\n'), lb2 = ws2 && (c2 == '\r' || c2 == '\n');
  const cp_t nn[2] = {'\n', '\n'}, nrn[3] = {'\n', '\r', '\n'}, rnn[3] = {'\r', '\n', '\n'}, rnrn[4] = {'\r', '\n', '\r', '\n'};
  const bool bl1 = lb1 && (tail_is(c, one, nn, 2) || tail_is(c, one, nrn, 3));
  const bool bl2 = lb2 && (head_is(c, two, nn, 2) || head_is(c, two, rnn, 3) || head_is(c, two, nrn, 3) || head_is(c, two, rnrn, 4));
  if (bl1 || bl2) return 5;
  if (lb1 || lb2) return 4;
  if (na1 && !ws1 && ws2) return 3;
  if (ws1 || ws2) return 2;
  if (na1 || na2) return 1;
  return 0;
}

ANX_HDS void cleanup_semantic_lossless(Ctx& c) {
  int64_t pointer = 1;
  while (pointer + 1 < (int64_t)c.nd && !c.overflow) {
    const uint32_t pt = (uint32_t)pointer;
    if (d_op(c, pt - 1) == '=' && d_op(c, pt + 1) == '=') {
      View eq1 = d_text(c, pt - 1), edit = d_text(c, pt), eq2 = d_text(c, pt + 1);
      const uint32_t co = common_suffix(c, eq1, edit);
      if (co) {
        const View cs = sub(edit, edit.n - co);
        eq1 = sub(eq1, 0, eq1.n - co);
        edit = concat(c, cs, sub(edit, 0, edit.n - co));
        eq2 = concat(c, cs, eq2);
      }
      View b1 = eq1, be = edit, b2 = eq2;
      int best = semantic_score(c, eq1, edit) + semantic_score(c, edit, eq2);
      while (edit.n && eq2.n && at(c, edit, 0) == at(c, eq2, 0) && !c.overflow) {
        eq1 = concat(c, eq1, sub(edit, 0, 1));
        edit = concat(c, sub(edit, 1), sub(eq2, 0, 1));
        eq2 = sub(eq2, 1);
        const int sc = semantic_score(c, eq1, edit) + semantic_score(c, edit, eq2);
        if (sc >= best) { best = sc; b1 = eq1; be = edit; b2 = eq2; }
      }
      if (c.overflow) return;
      if (!same(c, d_text(c, pt - 1), b1)) {
        if (b1.n) set_text(c, (uint32_t)pointer - 1, b1);
        else { d_erase(c, (uint32_t)pointer - 1, 1); --pointer; }
        set_text(c, (uint32_t)pointer, be);
        if (b2.n) set_text(c, (uint32_t)pointer + 1, b2);
        else { d_erase(c, (uint32_t)pointer + 1, 1); --pointer; }
      }
    }
    ++pointer;
  }
}

ANX_HDS void cleanup_semantic(Ctx& c) {
  bool changes = false;
  auto EQ = [&](uint32_t i) -> int32_t& { return reinterpret_cast<int32_t&>(W(c, c.v_off + i)); };  // stack of diff indices (bisect is done with the array)
  uint32_t neq = 0;
  bool have_last = false;
  View last_eq = empty(c);
  int64_t pointer = 0;
  uint32_t li1 = 0, ld1 = 0, li2 = 0, ld2 = 0;
  while (pointer < (int64_t)c.nd && !c.overflow) {
    const uint32_t op = d_op(c, (uint32_t)pointer);
    if (op == '=') {
      if (neq >= c.v_cap) { c.overflow = true; return; }
      EQ(neq++) = (int32_t)pointer;
      li1 = li2; li2 = 0; ld1 = ld2; ld2 = 0;
      last_eq = d_text(c, (uint32_t)pointer);
      have_last = true;
    } else {
      if (op == '+') li2 += d_len(c, (uint32_t)pointer);
      else ld2 += d_len(c, (uint32_t)pointer);
      const uint32_t m1 = li1 > ld1 ? li1 : ld1, m2 = li2 > ld2 ? li2 : ld2;
      if (have_last && last_eq.n && last_eq.n <= m1 && last_eq.n <= m2) {
        const int32_t e = EQ(neq - 1);
        d_insert(c, (uint32_t)e, '-', last_eq);
        if (c.overflow) return;
        set_op(c, (uint32_t)e + 1, '+');
        --neq;
        if (neq) --neq;
        pointer = neq ? EQ(neq - 1) : -1;
        li1 = ld1 = li2 = ld2 = 0;
        have_last = false;
        last_eq = empty(c);
        changes = true;
      }
    }
    ++pointer;
  }
  if (c.overflow) return;
  if (changes) cleanup_merge(c, 0);
  cleanup_semantic_lossless(c);
  uint32_t p = 1;
  while (p < c.nd && !c.overflow) {
    if (d_op(c, p - 1) == '-' && d_op(c, p) == '+') {
      const View deletion = d_text(c, p - 1), insertion = d_text(c, p);
      const uint32_t o1 = common_overlap(c, deletion, insertion), o2 = common_overlap(c, insertion, deletion);
      if (o1 >= o2) {
        if (2 * o1 >= deletion.n || 2 * o1 >= insertion.n) {
          d_insert(c, p, '=', sub(insertion, 0, o1));
          if (c.overflow) return;
          set_text(c, p - 1, sub(deletion, 0, deletion.n - o1));
          set_text(c, p + 1, sub(insertion, o1));
          ++p;
        }
      } else if (2 * o2 >= deletion.n || 2 * o2 >= insertion.n) {
        d_insert(c, p, '=', sub(deletion, 0, o2));
        if (c.overflow) return;
        set_op(c, p - 1, '+');
        set_text(c, p - 1, sub(insertion, 0, insertion.n - o2));
        set_op(c, p + 1, '-');
        set_text(c, p + 1, sub(deletion, o2));
        ++p;
      }
      ++p;
    }
    ++p;
  }
}

// shortest_edit_script(a, b): the diffs [0, nd) of the context (empty texts removed); false = the context overflowed
ANX_HDS bool edit_script(Ctx& c, const View& a, const View& b) {
  c.arena_used = 0;
  c.nd = 0;
  c.overflow = false;
  diff_main(c, a, b);
  if (!c.overflow) cleanup_semantic(c);
  if (!c.overflow) cleanup_merge(c, 0);
  if (c.overflow) return false;
  uint32_t w = 0;
  for (uint32_t i = 0; i < c.nd; ++i)
    if (d_len(c, i)) { if (w != i) d_move(c, w, i); ++w; }
  c.nd = w;
  return true;
}

ANX_HDS CharSet charset_of(const Ctx& c, const View& s) {
  CharSet cs;
  cs.w[0] = cs.w[1] = 0;
  cs.other = 0;
  for (uint32_t i = 0; i < s.n; ++i) {
    const cp_t ch = at(c, s, i);
    if (ch < 64) cs.w[0] |= 1ull << ch;
    else if (ch < 128) cs.w[1] |= 1ull << (ch & 63);
    else cs.other = 1;
  }
  return cs;
}

// found_in, src/confusables.rs:47-127, over the diffs [0, nd) of the context
ANX_HDS bool found_in(const Ctx& c, const Patterns& P, const FlatConf& cf) {
  const uint32_t l = cf.nops, nref = c.nd;
  uint32_t matches = 0;
  for (uint32_t i = 0; i < nref; ++i) {
    if (matches >= l) continue;
    const FlatOp& o = P.ops[cf.op_begin + matches];
    bool found = false;
    if (o.op == d_op(c, i)) {
      const View text = d_text(c, i);
      for (uint32_t k = 0; k < o.nopt; ++k) {
        const PView s = pmk(P.pool + P.opts[o.opt_begin + k].off, P.opts[o.opt_begin + k].len);
        bool ok;
        if (o.op != '=') ok = ends_with(c, text, s);
        else if (matches == 0 && matches == l - 1) ok = same(c, s, text);
        else if (matches == 0) ok = ends_with(c, text, s);
        else if (matches == l - 1) ok = starts_with(c, text, s);
        else ok = same(c, s, text);
        if (ok) { found = true; break; }
      }
    }
    if (!found) {
      matches = 0;
      if (cf.strictbegin) return false;
      continue;
    }
    if (++matches == l) return cf.strictend ? i == nref - 1 : true;
  }
  return false;
}

// A pattern can only be found in the edit script of (input -> candidate) if every instruction has an option that occurs in the
// string the instruction's text is taken from: a deletion's text is a piece of the input, an insertion's a piece of the
// candidate, an equality's a piece of both; with `$` the last diff is the end of the strings.  Necessary, not sufficient: it only
// decides whether the edit script is worth computing.
ANX_HDS bool occurs(const Ctx& c, const PView& opt, const View& s, const CharSet& cs) {
  if (opt.n == 1 && opt.p[0] < 128) return ((opt.p[0] < 64 ? cs.w[0] : cs.w[1]) >> (opt.p[0] & 63)) & 1ull;
  return opt.n == 0 ? true : find(c, s, opt) >= 0;
}
ANX_HDS bool may_match(const Ctx& c, const Patterns& P, const FlatConf& cf, const View& in, const CharSet& ins, const View& cand, const CharSet& cs) {
  for (uint32_t k = 0; k < cf.nops; ++k) {
    const FlatOp& o = P.ops[cf.op_begin + k];
    const bool tail = cf.strictend && k == cf.nops - 1 && o.op != '=';
    if (o.simple && !tail) {
      uint64_t h0 = o.bits[0], h1 = o.bits[1];
      if (o.op != '+') { h0 &= ins.w[0]; h1 &= ins.w[1]; }
      if (o.op != '-') { h0 &= cs.w[0]; h1 &= cs.w[1]; }
      if (!(h0 | h1)) return false;
      continue;
    }
    bool any = false;
    for (uint32_t j = 0; j < o.nopt; ++j) {
      const PView opt = pmk(P.pool + P.opts[o.opt_begin + j].off, P.opts[o.opt_begin + j].len);
      bool ok = (o.op == '+' || occurs(c, opt, in, ins)) && (o.op == '-' || occurs(c, opt, cand, cs));
      if (ok && tail) ok = ends_with(c, o.op == '+' ? cand : in, opt);  // the last diff ends the script: its text ends the string
      if (ok) { any = true; break; }
    }
    if (!any) return false;
  }
  return true;
}

// compute_confusable_weight (src/lib.rs:1733-1756) for one (input, candidate), both in the lane memory: the product of the weights
// of the patterns found in the edit script, 1.0 when no pattern can match (the script is then not computed).  false = the context
// overflowed.
ANX_HDS bool confusable_weight(Ctx& c, const Patterns& P, const View& in, const CharSet& ins, const View& cand, const CharSet& cs, double* weight) {
  *weight = 1.0;
  // the patterns that can match at all (bit j of `live`; with more than 64 patterns the later ones always count as live): only
  // those are looked for in the script -- may_match is a necessary condition of found_in
  uint64_t live = 0;
  bool any = false;
  for (uint32_t j = 0; j < P.nconf; ++j) {
    if (may_match(c, P, P.conf[j], in, ins, cand, cs)) {
      any = true;
      if (j < 64) live |= 1ull << j;
      else break;  // the rest is tested in the script
    }
  }
  if (!any) return true;
  if (!edit_script(c, in, cand)) return false;
  double w = 1.0;
  for (uint32_t j = 0; j < P.nconf; ++j)
    if ((j >= 64 || ((live >> j) & 1ull)) && found_in(c, P, P.conf[j])) w *= P.conf[j].weight;
  *weight = w;
  return true;
}
};  // Core<S>

}  // namespace cdiff
}  // namespace anx
