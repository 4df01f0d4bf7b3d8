// dgmi_tuning.h — the handful of launch parameters the measurement tools override (tools/*.py, one test).
// Read ONCE from the environment (DGMI_*), when the library is first asked for them, and afterwards changed only
// through dgmi_set_tuning (include/dgmi.h): no launch path calls getenv.  0 / -1 = the built-in choice.
#pragma once
#include <stdint.h>

namespace dgmi {

struct Tuning {
  int sliced_rows = 0;              // DGMI_SLICED_ROWS: destination rows per lane group of the XCD-local kernel
  int sliced_touch_lead = -1;       // DGMI_SLICED_PF: worker blocks between a toucher and what it touches (0: no touchers)
  int sliced_lpr = 0;               // DGMI_SLICED_LPR: lane-group width 8 / 16 / 32 / 64
  int sliced_no_off32 = 0;          // DGMI_NO_OFF32: 64-bit row addresses even where 32-bit offsets fit
  int64_t sliced_chunk_rows = 0;    // DGMI_SLICED_CHUNK_ROWS: destination rows per launch pair
  int64_t select_window_min = 0;    // DGMI_SELECT_WINDOW_MIN: shortest list that takes the window passes
  int select_narrow_window = 0;     // DGMI_SELECT_NARROW_WINDOW: a window that misses (forces the take-over path; test)
  int sort_plain_tiles = 0;         // DGMI_SORT_PLAIN_TILES: record sort with tile = blockIdx.x instead of the XCD-aware order
  int knn_screen_first = 0;         // DGMI_KNN_SCREEN_V1: the first (register-staged) 256 x 256 screen kernel instead of the LDS-DMA one (A/B tools)
  int64_t knn_pool_chunks = 0;      // DGMI_KNN_POOL_CHUNKS: chunks of the screen's record pool (a pool that runs out: the direct appends; test)
};

Tuning& tuning();  // dgmi_api.hip

}  // namespace dgmi
