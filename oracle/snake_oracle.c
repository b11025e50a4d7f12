/*
 * snake_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * State-machine restatement of SnakeGridMapping::SnakeGridIterator
 * (snake.h:23-80) and doMapping (snake.h:105-118).  PINNED: checked against
 * the reference header itself compiled into oracle/_ref/libsnake_ref.so
 * (tests/test_oracle_snake.py).
 */
#include "oracle_util.h"

void pcco_snake_perm(int w, int h, int32_t *perm) {
  /* iterator state, snake.h:27-35 */
  int w_pos = 0, h_pos = 0, mb_w = 0, mb_h = 0, to_right = 1;
  for (int i = 0; i < w * h; i++) {
    /* operator++(int), snake.h:67-71: value first, then advance */
    perm[i] = w_pos + (h_pos + 8 * mb_h) * w + mb_w * 8;
    /* updatePos(), snake.h:46-64 */
    if (to_right) w_pos++; else w_pos--;
    if (((w_pos % 8 == 0) && to_right) || w_pos < 0) {
      h_pos++;
      to_right = !to_right;
      w_pos = to_right ? 0 : 7;
      if (h_pos % 8 == 0 || (h_pos + mb_h * 8 == h)) {
        h_pos = 0;
        mb_w++;
        if (mb_w % (w / 8) == 0) {
          mb_w = 0;
          mb_h++;
        }
      }
    }
  }
}
