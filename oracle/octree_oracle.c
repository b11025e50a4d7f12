/*
 * octree_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Pointer-octree restatement of the geometry part of the hot path, with the
 * reference's own cost structure (per-point insertion, pointer chasing, one
 * allocation per node, depth-first serialisation with a per-leaf gather):
 *
 *   P1 addPointsFromInputCloud / addPointIdx      (called at impl.hpp:99)
 *   P2 adoptBoundingBoxToPoint + getKeyBitSize    (inside P1)
 *   P3 genOctreeKeyforPoint                       (inside P1)
 *   P4 createLeafRecursive + addPointIndex        (inside P1)
 *   P5 serializeTree / serializeTreeRecursive / getBranchBitPattern (impl.hpp:166)
 *   C2 serializeTreeCallback                      (impl.hpp:1509-1578)
 *   P6 ColorCoding::encodeAverageOfPoints         (impl.hpp:1548-1550)
 *   C4 PointCodingV2::encodePoint                 (ptv2.h:83-97)
 *
 * P1-P6 live in PCL (1.8.1-1.10.0), which is neither vendored in
 * /root/reference nor installed: "parity unpinned" (SURVEY.md 8c).  The
 * restatement follows PCL 1.10.0 octree_pointcloud.hpp / octree2buf_base.hpp /
 * color_coding.h semantics as recorded in SURVEY.md section 8(a).
 */
#include "oracle_util.h"
#include "octree_oracle.h"
#include <float.h>
#include <math.h>

typedef struct obranch { void *child[8]; } obranch;
typedef struct oleaf { int *idx; int n, cap; } oleaf;

struct pcco_octree {
  double res;
  double min[3], max[3];
  int bbox_defined;
  int too_deep;         /* the box would need a 33rd level: PCL's own `1 << octree_depth_` is undefined from there on (it is what
                           this file restates), so the restatement stops and says so instead of running away with it */
  unsigned depth;       /* octree_depth_ */
  obranch *root;        /* root_node_ (always a branch) */
  uint64_t leaf_count, branch_count, object_count;
};

pcco_octree *pcco_octree_new(double resolution) {
  pcco_octree *t = (pcco_octree *)calloc(1, sizeof(*t));
  t->res = resolution;
  t->root = (obranch *)calloc(1, sizeof(obranch));
  t->branch_count = 1;
  return t;
}

static void free_rec(void *node, unsigned level, unsigned depth) {
  if (!node) return;
  if (level == depth) {
    oleaf *l = (oleaf *)node;
    free(l->idx);
    free(l);
    return;
  }
  obranch *b = (obranch *)node;
  for (int c = 0; c < 8; c++) free_rec(b->child[c], level + 1, depth);
  free(b);
}

void pcco_octree_free(pcco_octree *t) {
  if (!t) return;
  free_rec(t->root, 0, t->depth ? t->depth : 1);
  free(t);
}

/* P2: OctreePointCloud::getKeyBitSize, first-point branch only (the tree is
 * empty whenever the reference reaches it: deleteTree() at impl.hpp:90). */
static void get_key_bit_size(pcco_octree *t) {
  const float min_value = FLT_EPSILON;
  unsigned max_key[3];
  for (int a = 0; a < 3; a++)
    max_key[a] = (unsigned)ceil((t->max[a] - t->min[a] - min_value) / t->res);
  unsigned max_voxels = max_key[0];
  if (max_key[1] > max_voxels) max_voxels = max_key[1];
  if (max_key[2] > max_voxels) max_voxels = max_key[2];
  if (max_voxels < 2) max_voxels = 2;
  unsigned d = (unsigned)ceil(log2((double)max_voxels) - min_value);
  if (d > 32) d = 32;
  t->depth = d;
  double side = (double)(1u << t->depth) * t->res;
  if (t->leaf_count == 0) {
    for (int a = 0; a < 3; a++) {
      double over = (side - (t->max[a] - t->min[a])) / 2.0;
      if (over > min_value) {
        t->min[a] -= over;
        t->max[a] += over;
      }
    }
  } else {
    for (int a = 0; a < 3; a++) t->max[a] = t->min[a] + side;
  }
}

/* P2: OctreePointCloud::adoptBoundingBoxToPoint. */
static void adopt_bbox(pcco_octree *t, const float p[3]) {
  const float min_value = FLT_EPSILON;
  for (;;) {
    int lo[3], up[3], any = 0;
    for (int a = 0; a < 3; a++) {
      lo[a] = ((double)p[a] < t->min[a]);
      up[a] = ((double)p[a] >= t->max[a]);
      any |= lo[a] | up[a];
    }
    if (!(any || !t->bbox_defined)) break;
    if (t->bbox_defined) {
      if (t->depth >= 32) { t->too_deep = 1; return; }
      /* grow: the old root becomes child ((!upX)<<2 | (!upY)<<1 | !upZ) of a new root */
      unsigned child_idx = (unsigned)(((!up[0]) << 2) | ((!up[1]) << 1) | (!up[2]));
      obranch *nr = (obranch *)calloc(1, sizeof(obranch));
      t->branch_count++;
      nr->child[child_idx] = t->root;
      t->root = nr;
      double side = (double)(1u << t->depth) * t->res;
      for (int a = 0; a < 3; a++)
        if (!up[a]) t->min[a] -= side;
      t->depth++;
      side = (double)(1u << t->depth) * t->res - min_value;
      for (int a = 0; a < 3; a++) t->max[a] = t->min[a] + side;
    } else {
      for (int a = 0; a < 3; a++) {
        t->min[a] = (double)p[a] - t->res / 2;
        t->max[a] = (double)p[a] + t->res / 2;
      }
      get_key_bit_size(t);
      t->bbox_defined = 1;
    }
  }
}

/* P1/P3/P4: OctreePointCloudCompression::addPointIdx -> OctreePointCloud::addPointIdx. */
static void add_point_idx(pcco_octree *t, const pcco_point *pts, int i) {
  t->object_count++;
  float p[3] = {pts[i].x, pts[i].y, pts[i].z};
  adopt_bbox(t, p);
  if (t->too_deep) return;
  /* genOctreeKeyforPoint */
  unsigned key[3];
  for (int a = 0; a < 3; a++) key[a] = (unsigned)(((double)p[a] - t->min[a]) / t->res);
  /* createLeafRecursive: MSB -> LSB */
  obranch *b = t->root;
  for (unsigned level = 0; level < t->depth; level++) {
    unsigned mask = 1u << (t->depth - 1 - level);
    unsigned c = ((!!(key[0] & mask)) << 2) | ((!!(key[1] & mask)) << 1) | (!!(key[2] & mask));
    if (level + 1 == t->depth) {
      oleaf *l = (oleaf *)b->child[c];
      if (!l) {
        l = (oleaf *)calloc(1, sizeof(oleaf));
        b->child[c] = l;
        t->leaf_count++;
      }
      if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 1;
        l->idx = (int *)realloc(l->idx, sizeof(int) * (size_t)l->cap);
      }
      l->idx[l->n++] = i; /* OctreeContainerPointIndices::addPointIndex */
    } else {
      obranch *nb = (obranch *)b->child[c];
      if (!nb) {
        nb = (obranch *)calloc(1, sizeof(obranch));
        b->child[c] = nb;
        t->branch_count++;
      }
      b = nb;
    }
  }
}

void pcco_octree_add_points(pcco_octree *t, const pcco_point *pts, size_t n) {
  for (size_t i = 0; i < n; i++) {
    /* pcl::isFinite(PointXYZRGB): x, y and z finite */
    if (isfinite(pts[i].x) && isfinite(pts[i].y) && isfinite(pts[i].z))
      add_point_idx(t, pts, (int)i);
    if (t->too_deep) return;
  }
}
int pcco_octree_too_deep(const pcco_octree *t) { return t->too_deep; }

uint64_t pcco_octree_leaf_count(const pcco_octree *t) { return t->leaf_count; }
uint64_t pcco_octree_object_count(const pcco_octree *t) { return t->object_count; }
unsigned pcco_octree_depth(const pcco_octree *t) { return t->depth; }
void pcco_octree_bbox(const pcco_octree *t, double bb[6]) {
  for (int a = 0; a < 3; a++) { bb[a] = t->min[a]; bb[3 + a] = t->max[a]; }
}

/* ---- serialisation ---- */

typedef struct {
  const pcco_octree *t;
  const pcco_point *pts;
  const pcco_params *prm;
  int cloud_with_color;
  pcco_frame *f;
  size_t leaf_i;
} ser_ctx;

/* C2: serializeTreeCallback, voxel-grid branch (impl.hpp:1542-1577). */
static void leaf_callback(ser_ctx *s, const oleaf *l, const unsigned key[3]) {
  const pcco_octree *t = s->t;
  pcco_frame *f = s->f;
  double lc[3];
  for (int a = 0; a < 3; a++) lc[a] = (double)key[a] * t->res + t->min[a]; /* impl.hpp:1519-1521 */

  pcco_point c; /* default-constructed pcl::PointXYZRGB */
  memset(&c, 0, sizeof(c));
  c.w = 1.0f;
  uint8_t r = 0, g = 0, b = 0;

  if (s->cloud_with_color) {
    /* P6: ColorCoding::encodeAverageOfPoints -- note "avgRed" reads byte 0 (= blue) */
    unsigned s0 = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < l->n; k++) {
      uint32_t w = s->pts[l->idx[k]].rgba;
      s0 += (w >> 0) & 0xFF;
      s1 += (w >> 8) & 0xFF;
      s2 += (w >> 16) & 0xFF;
    }
    if (l->n > 1) {
      s0 /= (unsigned)l->n;
      s1 /= (unsigned)l->n;
      s2 /= (unsigned)l->n;
    }
    /* colorBitReduction_: only the PCL colour coder (type 0) ever gets setBitDepth */
    unsigned red = s->prm->color_coding_type ? 0u : (unsigned)(8 - s->prm->color_bit_resolution);
    s0 >>= red; s1 >>= red; s2 >>= red;
    buf_put(&f->bgr, (uint8_t)s0);
    buf_put(&f->bgr, (uint8_t)s1);
    buf_put(&f->bgr, (uint8_t)s2);
    r = (uint8_t)s2; g = (uint8_t)s1; b = (uint8_t)s0; /* impl.hpp:1554-1556 */
  }
  if (!s->prm->do_voxel_centroid) {
    c.x = (float)(lc[0] + 0.5 * t->res); /* impl.hpp:1560-1562 */
    c.y = (float)(lc[1] + 0.5 * t->res);
    c.z = (float)(lc[2] + 0.5 * t->res);
  } else {
    /* pcl::compute3DCentroid<PointT,float>: float accumulate in index order, then divide */
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = 0; k < l->n; k++) {
      const pcco_point *q = &s->pts[l->idx[k]];
      sx += q->x; sy += q->y; sz += q->z;
    }
    float cnt = (float)l->n;
    c.x = sx / cnt; c.y = sy / cnt; c.z = sz / cnt;
    /* C4: PointCodingV2::encodePoint, precision = PointCoding default 0.001f */
    const float prec = 0.001f;
    float cc[3] = {c.x, c.y, c.z};
    for (int a = 0; a < 3; a++) {
      int d = (int)(((double)cc[a] - lc[a]) / (double)prec);
      if (d > 127) d = 127;
      if (d < -127) d = -127;
      buf_put(&f->centroid_bytes, (uint8_t)d);
    }
  }
  c.rgba = (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16) | (0xFFu << 24);
  f->simplified[s->leaf_i] = c;
  f->leaf_keys[3 * s->leaf_i + 0] = key[0];
  f->leaf_keys[3 * s->leaf_i + 1] = key[1];
  f->leaf_keys[3 * s->leaf_i + 2] = key[2];
  f->leaf_counts[s->leaf_i] = (uint32_t)l->n;
  s->leaf_i++;
}

/* P5: Octree2BufBase::serializeTreeRecursive with do_XOR = false. */
static void serialize_rec(ser_ctx *s, const obranch *b, unsigned level, unsigned key[3]) {
  uint8_t bits = 0; /* getBranchBitPattern */
  for (int c = 0; c < 8; c++) bits |= (uint8_t)((!!b->child[c]) << c);
  buf_put(&s->f->occupancy, bits);
  for (unsigned c = 0; c < 8; c++) {
    if (!b->child[c]) continue;
    /* OctreeKey::pushBranch */
    key[0] = (key[0] << 1) | (!!(c & 4));
    key[1] = (key[1] << 1) | (!!(c & 2));
    key[2] = (key[2] << 1) | (!!(c & 1));
    if (level + 1 == s->t->depth)
      leaf_callback(s, (const oleaf *)b->child[c], key);
    else
      serialize_rec(s, (const obranch *)b->child[c], level + 1, key);
    key[0] >>= 1; key[1] >>= 1; key[2] >>= 1; /* popBranch */
  }
}

void pcco_octree_serialize(const pcco_octree *t, const pcco_point *pts, const pcco_params *prm,
                           int cloud_with_color, pcco_frame *f) {
  ser_ctx s;
  s.t = t; s.pts = pts; s.prm = prm; s.cloud_with_color = cloud_with_color; s.f = f; s.leaf_i = 0;
  size_t L = (size_t)t->leaf_count;
  f->leaf_keys = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (L ? L : 1));
  f->leaf_counts = (uint32_t *)malloc(sizeof(uint32_t) * (L ? L : 1));
  f->simplified = (pcco_point *)malloc(sizeof(pcco_point) * (L ? L : 1));
  unsigned key[3] = {0, 0, 0};
  serialize_rec(&s, t->root, 0, key);
  f->n_leaves = L;
  f->n_branches = f->occupancy.len;
  f->depth = t->depth;
  f->n_points_in = t->object_count;
  pcco_octree_bbox(t, f->bbox);
}

/* ---- trees of the delta (inter-frame) path: the box is defined before the points are added ----
 * OctreePointCloud::defineBoundingBox(min, max) then addPointsFromInputCloud (impl.hpp:340-342, 426-428).
 * Returns the leaves in depth-first order: key triples, point counts and the concatenated point index lists
 * (each list in ascending point index).  The caller frees the three arrays. */
typedef struct { uint32_t *keys, *counts; int *indices; size_t li, pi; unsigned depth; } leaf_dump;

static void dump_rec(const obranch *b, unsigned level, unsigned depth, unsigned key[3], leaf_dump *d) {
  for (unsigned c = 0; c < 8; c++) {
    if (!b->child[c]) continue;
    key[0] = (key[0] << 1) | (!!(c & 4));
    key[1] = (key[1] << 1) | (!!(c & 2));
    key[2] = (key[2] << 1) | (!!(c & 1));
    if (level + 1 == depth) {
      const oleaf *l = (const oleaf *)b->child[c];
      d->keys[3 * d->li] = key[0]; d->keys[3 * d->li + 1] = key[1]; d->keys[3 * d->li + 2] = key[2];
      d->counts[d->li] = (uint32_t)l->n;
      memcpy(d->indices + d->pi, l->idx, sizeof(int) * (size_t)l->n);
      d->pi += (size_t)l->n;
      d->li++;
    } else {
      dump_rec((const obranch *)b->child[c], level + 1, depth, key, d);
    }
    key[0] >>= 1; key[1] >>= 1; key[2] >>= 1;
  }
}

int pcco_tree_with_defined_box(const pcco_point *pts, size_t n, double res, const double box[6],
                               uint32_t **keys, uint32_t **counts, int **indices, uint64_t *n_leaves,
                               double bbox_out[6], unsigned *depth) {
  pcco_octree *t = pcco_octree_new(res);
  for (int a = 0; a < 3; a++) { t->min[a] = box[a]; t->max[a] = box[3 + a]; }
  get_key_bit_size(t);   /* defineBoundingBox: leaf_count_ == 0 here */
  t->bbox_defined = 1;
  pcco_octree_add_points(t, pts, n);
  leaf_dump d;
  size_t L = (size_t)t->leaf_count;
  d.keys = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (L ? L : 1));
  d.counts = (uint32_t *)malloc(sizeof(uint32_t) * (L ? L : 1));
  d.indices = (int *)malloc(sizeof(int) * ((size_t)t->object_count ? (size_t)t->object_count : 1));
  d.li = d.pi = 0;
  unsigned key[3] = {0, 0, 0};
  if (t->depth) dump_rec(t->root, 0, t->depth, key, &d);
  *keys = d.keys; *counts = d.counts; *indices = d.indices;
  *n_leaves = L;
  *depth = t->depth;
  pcco_octree_bbox(t, bbox_out);
  pcco_octree_free(t);
  return 0;
}
