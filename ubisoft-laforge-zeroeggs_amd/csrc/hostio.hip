// Host-side text formatting of the BVH motion block (reference ZEGGS/anim/bvh.py save(): one row per frame, "%f" per channel).
// Pure host code (no kernel): the 30-minute clip of generate.py's long-form use is 108 000 rows x 228 channels = 24.6 M
// numbers, which numpy.savetxt formats at ~1 us each in the interpreter; snprintf does the same correctly-rounded "%f" at
// ~0.1 us.  File I/O stays on the host as north_star asks; the channel values come from zeggs_pose_to_bvh (device).
#include <stdio.h>
#include <vector>

#include "../../include/zeggs_hip.h"
#include "common.h"

extern "C" int zeggs_write_table_text(const char* path, int append, const double* table, long rows, int cols) {
  ZCHECK(path && table && rows >= 0 && cols > 0, "write_table_text: bad arguments");
  FILE* f = fopen(path, append ? "a" : "w");
  ZCHECK(f != nullptr, "write_table_text: cannot open %s", path);
  std::vector<char> buf((size_t)cols * 330 + 8);      // "%f" of a double is at most 1 + 309 + 1 + 6 characters
  static char big[1 << 20];
  setvbuf(f, big, _IOFBF, sizeof(big));
  for (long r = 0; r < rows; ++r) {
    char* p = buf.data();
    const double* row = table + r * cols;
    for (int c = 0; c < cols; ++c) {
      p += snprintf(p, 328, "%f", row[c]);
      *p++ = ' ';
    }
    *p++ = '\n';
    if (fwrite(buf.data(), 1, (size_t)(p - buf.data()), f) != (size_t)(p - buf.data())) {
      fclose(f);
      zeggs_set_error("write_table_text: short write to %s", path);
      return -1;
    }
  }
  ZCHECK(fclose(f) == 0, "write_table_text: close failed for %s", path);
  return 0;
}
