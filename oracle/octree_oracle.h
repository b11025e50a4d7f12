/* octree_oracle.h -- CPU ORACLE (test infrastructure): pointer-octree internals. */
#ifndef PCC_OCTREE_ORACLE_H
#define PCC_OCTREE_ORACLE_H
#include "pcc_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct pcco_octree pcco_octree;
pcco_octree *pcco_octree_new(double resolution);
void pcco_octree_free(pcco_octree *t);
void pcco_octree_add_points(pcco_octree *t, const pcco_point *pts, size_t n);
uint64_t pcco_octree_leaf_count(const pcco_octree *t);
uint64_t pcco_octree_object_count(const pcco_octree *t);
unsigned pcco_octree_depth(const pcco_octree *t);
int pcco_octree_too_deep(const pcco_octree *t);  /* the adaptive box asked for a 33rd level: nothing was coded */
void pcco_octree_bbox(const pcco_octree *t, double bb[6]);
void pcco_octree_serialize(const pcco_octree *t, const pcco_point *pts, const pcco_params *prm,
                           int cloud_with_color, pcco_frame *f);
/* delta path: tree with defineBoundingBox() before the points (impl.hpp:340-342, 426-428); leaves in DFS order */
int pcco_tree_with_defined_box(const pcco_point *pts, size_t n, double res, const double box[6],
                               uint32_t **keys, uint32_t **counts, int **indices, uint64_t *n_leaves,
                               double bbox_out[6], unsigned *depth);
#ifdef __cplusplus
}
#endif
#endif
