// ref_snake_driver.cpp -- CPU ORACLE support (test infrastructure, not product code).
//
// Thin extern "C" driver around the REFERENCE's own SnakeGridMapping header,
// compiled from where it lies under /root/reference (never copied into this
// repo).  snake_grid_mapping.h is the one file of the hot path that is
// self-contained (needs only <vector>/<stdint.h>), so it can be built here
// without PCL; everything else in the reference needs PCL/Boost/Eigen and is
// unbuildable in this image.  Output: oracle/_ref/libsnake_ref.so.
#include <cstdint>   // the header names std::uint8_t but only includes <stdint.h>; the reference gets <cstdint> transitively
#include <cstring>
#include <pcl/cloud_codec_v2/snake_grid_mapping.h>

extern "C" {

// doMapping (snake.h:105-118) on a 3*w*h byte vector.
void ref_snake_do_mapping(int w, int h, const uint8_t* in, uint8_t* out) {
  std::vector<char> v(in, in + 3 * (size_t)w * h);
  pcl::octree::SnakeGridMapping<char, uint8_t> m(w, h);
  std::vector<uint8_t>& r = m.doMapping(v);
  std::memcpy(out, r.data(), r.size());
}

// undoSnakeGridMapping (snake.h:123-137).
void ref_snake_undo_mapping(int w, int h, const uint8_t* in, uint8_t* out) {
  std::vector<uint8_t> v(in, in + 3 * (size_t)w * h);
  pcl::octree::SnakeGridMapping<uint8_t, char> m(w, h);
  std::vector<char>& r = m.undoSnakeGridMapping(v);
  std::memcpy(out, r.data(), r.size());
}

// The iterator's position sequence (snake.h:67-71), i -> pixel index.
void ref_snake_perm(int w, int h, int32_t* perm) {
  pcl::octree::SnakeGridMapping<uint8_t, uint8_t>::SnakeGridIterator it(w, h);
  for (int i = 0; i < w * h; i++) perm[i] = it++;
}
}
