// scan_i8.hip -- int8 slab variant of the cosine scan (placeholder until the kernel lands).
#include "scan.h"
namespace crs {
int scan_launch_i8(const ScanArgs&, int, int, hipStream_t) { return -1; }
}  // namespace crs
