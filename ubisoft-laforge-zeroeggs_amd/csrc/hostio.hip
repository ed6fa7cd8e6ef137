// Host-side text formatting of the BVH motion block (reference ZEGGS/anim/bvh.py save(): one row per frame, "%f" per channel).
// Pure host code (no kernel): the 30-minute clip of generate.py's long-form use is 108 000 rows x 228 channels = 24.6 M
// numbers, which numpy.savetxt formats at ~1 us each in the interpreter; snprintf does the same correctly-rounded "%f" at
// ~0.15 us, and the rows are independent, so chunks of rows are formatted by a few host threads into their own buffers and
// written in order.  File I/O stays on the host as north_star asks; the channel values come from zeggs_pose_to_bvh (device).
#include <stdio.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/zeggs_hip.h"
#include "common.h"

namespace {
// rows [r0, r1) of `table` -> text ("%f" + ' ' per number, '\n' per row), appended to out
void format_rows(const double* table, long r0, long r1, int cols, std::string* out) {
  std::vector<char> line((size_t)cols * 330 + 8);      // "%f" of a double is at most 1 + 309 + 1 + 6 characters
  out->reserve((size_t)(r1 - r0) * cols * 11);
  for (long r = r0; r < r1; ++r) {
    char* p = line.data();
    const double* row = table + r * cols;
    for (int c = 0; c < cols; ++c) {
      p += snprintf(p, 328, "%f", row[c]);
      *p++ = ' ';
    }
    *p++ = '\n';
    out->append(line.data(), (size_t)(p - line.data()));
  }
}
}  // namespace

extern "C" int zeggs_write_table_text(const char* path, int append, const double* table, long rows, int cols) {
  ZCHECK(path && table && rows >= 0 && cols > 0, "write_table_text: bad arguments");
  FILE* f = fopen(path, append ? "a" : "w");
  ZCHECK(f != nullptr, "write_table_text: cannot open %s", path);
  // a few MB of text per chunk; at most 16 formatting threads, none for small tables
  const long per = 2048;
  long nchunks = (rows + per - 1) / per;
  unsigned hw = std::thread::hardware_concurrency();
  int nthr = (int)(hw ? (hw < 16 ? hw : 16) : 4);
  if ((long)nthr > nchunks) nthr = (int)(nchunks > 0 ? nchunks : 1);
  bool ok = true;
  for (long c0 = 0; c0 < nchunks && ok; c0 += nthr) {      // waves of nthr chunks: bounded memory, rows written in order
    const int n = (int)((nchunks - c0) < nthr ? (nchunks - c0) : nthr);
    std::vector<std::string> text((size_t)n);
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) {
      const long r0 = (c0 + i) * per, r1 = r0 + per < rows ? r0 + per : rows;
      th.emplace_back(format_rows, table, r0, r1, cols, &text[(size_t)i]);
    }
    {
      const long r0 = c0 * per, r1 = r0 + per < rows ? r0 + per : rows;
      format_rows(table, r0, r1, cols, &text[0]);
    }
    for (auto& t : th) t.join();
    for (int i = 0; i < n && ok; ++i) ok = fwrite(text[(size_t)i].data(), 1, text[(size_t)i].size(), f) == text[(size_t)i].size();
  }
  if (!ok) {
    fclose(f);
    zeggs_set_error("write_table_text: short write to %s", path);
    return -1;
  }
  ZCHECK(fclose(f) == 0, "write_table_text: close failed for %s", path);
  return 0;
}

// The formatting half alone: rows x cols HOST doubles -> text in the caller's buffer (same "%f " per number, '\n' per row).
// Single-threaded and re-entrant: generate_gesture() calls it from several host threads on row blocks of the chunks a running
// decode has already delivered, and writes the blocks in order itself.  *written = bytes produced; -1 if `cap` is too small
// (cap >= rows * (cols * 24 + 1) always suffices for |values| < 1e15).
extern "C" int zeggs_format_table_text(const double* table, long rows, int cols, char* out, size_t cap, size_t* written) {
  ZCHECK(table && out && written && rows >= 0 && cols > 0, "format_table_text: bad arguments");
  char* p = out;
  char* const end = out + cap;
  for (long r = 0; r < rows; ++r) {
    const double* row = table + r * cols;
    for (int c = 0; c < cols; ++c) {
      ZCHECK(end - p > 330, "format_table_text: buffer too small (%zu bytes for %ld x %d)", cap, rows, cols);
      p += snprintf(p, 328, "%f", row[c]);
      *p++ = ' ';
    }
    *p++ = '\n';
  }
  *written = (size_t)(p - out);
  return 0;
}

// The reading half (bvh.load, ZEGGS/anim/bvh.py: the MOTION block is rows of whitespace-separated numbers): `text` [len] bytes ->
// rows x cols HOST doubles, correctly rounded (strtod, as Python's float()).  The 30-minute-capable exemplar of configs[4] is
// 7 200 rows x 228 numbers = 11 MB of text; numpy.loadtxt needs 76 ms for it, this ~8 ms: rows are independent, so the buffer is
// cut at line ends and parsed by a few host threads.  Returns -1 when the text does not hold exactly rows x cols numbers
// (the caller falls back to its own parser and reports the malformed file there).
namespace {
// numbers of the lines [p, end) -> out; stops after `want` numbers; returns how many were read (or -1 on a token that is no number)
long parse_span(const char* p, const char* end, double* out, long want) {
  long n = 0;
  while (p < end) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    if (p >= end) break;
    // fast path for plain decimals ("%f" output: sign, digits, '.', digits): mantissa < 2^53 and at most 22 decimals make
    // mantissa / 10^decimals ONE correctly rounded division of two exact doubles -- the value strtod returns; anything else
    // (exponents, inf / nan, longer mantissas) goes to strtod
    const char* q = p;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = *q == '-'; ++q; }
    unsigned long long m = 0;
    int nd = 0, dec = 0;
    while (q < end && *q >= '0' && *q <= '9' && nd < 18) { m = m * 10 + (unsigned)(*q - '0'); ++q; ++nd; }
    if (q < end && *q == '.') {
      ++q;
      while (q < end && *q >= '0' && *q <= '9' && nd < 18) { m = m * 10 + (unsigned)(*q - '0'); ++q; ++nd; ++dec; }
    }
    const bool term = q >= end || *q == ' ' || *q == '\t' || *q == '\n' || *q == '\r';
    double v;
    if (term && nd > 0 && m < (1ULL << 53)) {
      static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17,
                                   1e18};
      v = (double)m / p10[dec];
      if (neg) v = -v;
    } else {
      char* e = nullptr;
      v = strtod(p, &e);
      if (e == p) return -1;
      q = e;
    }
    if (n < want) out[n] = v;
    ++n;
    p = q;
  }
  return n;
}
}  // namespace
extern "C" int zeggs_parse_table_text(const char* text, size_t len, double* table, long rows, int cols) {
  ZCHECK(text && table && rows >= 0 && cols > 0, "parse_table_text: bad arguments");
  ZCHECK(len > 0 && text[len] == 0, "parse_table_text: the buffer must be NUL-terminated behind `len` bytes (strtod)");
  unsigned hw = std::thread::hardware_concurrency();
  int nthr = (int)(hw ? (hw < 16 ? hw : 16) : 4);
  if ((long)nthr > rows / 256) nthr = (int)(rows / 256 > 0 ? rows / 256 : 1);
  // byte ranges that start right behind a line end
  std::vector<size_t> lo((size_t)nthr + 1);
  for (int t = 0; t <= nthr; ++t) {
    size_t p = len * (size_t)t / (size_t)nthr;
    if (t > 0 && t < nthr) {
      const void* nl = memchr(text + p, '\n', len - p);
      p = nl ? (size_t)((const char*)nl - text) + 1 : len;
    }
    lo[(size_t)t] = t == nthr ? len : p;
  }
  auto blank = [](const char* a, const char* b) {
    for (; a < b; ++a) if (*a != ' ' && *a != '\t' && *a != '\r') return false;
    return true;
  };
  // pass 1: non-empty lines per range; pass 2: every such line is one row of exactly `cols` numbers
  std::vector<long> cnt((size_t)nthr, 0), bad((size_t)nthr, 0);
  auto walk = [&](int t, long row0, bool parse) {
    const char* p = text + lo[(size_t)t];
    const char* const end = text + lo[(size_t)t + 1];
    long r = row0;
    while (p < end) {
      const void* nl = memchr(p, '\n', (size_t)(end - p));
      const char* e = nl ? (const char*)nl : end;
      if (!blank(p, e)) {
        if (parse && r < rows && parse_span(p, e, table + r * cols, cols) != cols) { bad[(size_t)t] = 1; return; }
        ++r;
      }
      p = e + 1;
    }
    if (!parse) cnt[(size_t)t] = r - row0;
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; ++t) th.emplace_back(walk, t, 0L, false);
    walk(0, 0L, false);
    for (auto& x : th) x.join();
  }
  long total = 0;
  std::vector<long> first((size_t)nthr, 0);
  for (int t = 0; t < nthr; ++t) { first[(size_t)t] = total; total += cnt[(size_t)t]; }
  ZCHECK(total == rows, "parse_table_text: %ld non-empty lines, %ld rows expected", total, rows);
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; ++t) th.emplace_back(walk, t, first[(size_t)t], true);
    walk(0, 0L, true);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < nthr; ++t) ZCHECK(bad[(size_t)t] == 0, "parse_table_text: a row does not hold exactly %d numbers", cols);
  return 0;
}
