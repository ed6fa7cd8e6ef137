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
