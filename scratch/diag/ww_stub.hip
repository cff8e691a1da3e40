// The stamped diagnostic build (ww_stamp.sh) links only the Winograd weight-gradient kernel: the slab reduction it would chain to
// (csrc/wgrad.hip) is not part of the measurement.
extern "C" int sqd_wgrad_reduce_launch(const float*, float*, float*, int, long long, int, int, int, void*) { return 2; }
