// CPU check of the hierarchy builder's invariants (raytracer.c_amd/csrc/bvh_build.h): compiled by g++ in
// tests/test_host.py::test_hierarchy_builder_invariants, no GPU.  bvh_check() builds the tree over n triangles
// (9 doubles each: v0, e1, e2) and returns 0 when every invariant the device walk relies on holds:
//   1 order is a permutation;  2 the leaves tile [0, n) in order, each 0 < count <= PT_BVH_LEAF (an empty second child only at
//   the root of a one-leaf mesh);  3 depth <= the least depth such leaves allow (the LDS stack is sized by it) and equals the
//   longest path;  4 every stored child box is exactly the union of its subtree's triangle boxes;  5 two builds agree.
// out[0] = depth, out[1] = max_depth, out[2] = nodes, out[3] = leaves, out[4] = 1 when the surface-area tree was kept;
// *cost = the builder's own tree_cost() (expected instructions of a walk per ray that meets the root's box).
#include <cstdio>
#include "bvh_build.h"

namespace
{
struct Walk
{
  const BvhBuild &b;
  uint32_t next_first = 0, leaves = 0;
  int longest = 0, err = 0;
  double cost = 0;
  // returns the triangle count of the subtree; box = the union of its triangles' boxes
  uint32_t visit(uint32_t ref, double *box, int level, bool root_child1)
  {
    for (int k = 0; k < 3; k++)
    {
      box[k] = 1e300;
      box[3 + k] = -1e300;
    }
    if (ref & PT_BVH_LEAF_FLAG)
    {
      const uint32_t first = (ref & ~PT_BVH_LEAF_FLAG) >> PT_BVH_COUNT_BITS, count = ref & ((1u << PT_BVH_COUNT_BITS) - 1u);
      if (count == 0)
      {
        if (!root_child1)
          err = err ? err : 20; // an empty leaf anywhere but the one-leaf mesh's second child
        return 0;
      }
      if (count > PT_BVH_LEAF || first != next_first)
        err = err ? err : 21;
      next_first = first + count;
      leaves++;
      for (uint32_t i = first; i < first + count && i < b.order.size(); i++)
        for (int k = 0; k < 3; k++)
        {
          box[k] = std::fmin(box[k], b.lo[3 * b.order[i] + k]);
          box[3 + k] = std::fmax(box[3 + k], b.hi[3 * b.order[i] + k]);
        }
      return count;
    }
    if ((size_t)ref * PT_BVH_SRC_DOUBLES >= b.nodes.size())
    {
      err = err ? err : 22;
      return 0;
    }
    longest = std::max(longest, level + 1);
    const double *n = &b.nodes[(size_t)ref * PT_BVH_SRC_DOUBLES];
    uint32_t refs[2];
    memcpy(refs, n + 12, sizeof refs);
    uint32_t total = 0;
    for (int c = 0; c < 2; c++)
    {
      double cb[6];
      const uint32_t cnt = visit(refs[c], cb, level + 1, level == 0 && c == 1 && b.nodes.size() == PT_BVH_SRC_DOUBLES);
      if (cnt != 0)
      {
        for (int k = 0; k < 6; k++)
          if (!(cb[k] == n[6 * c + k])) // (numerically: a union formed in another order may hold -0.0 where this one has +0.0)
            err = err ? err : 23;       // the stored box is not the union of the subtree's boxes
        cost += BvhBuild::half_area(cb, cb + 3) * cnt;
      }
      for (int k = 0; k < 3; k++)
      {
        box[k] = std::fmin(box[k], cb[k]);
        box[3 + k] = std::fmax(box[3 + k], cb[3 + k]);
      }
      total += cnt;
    }
    return total;
  }
};

void build(BvhBuild &b, const double *tgeom, uint32_t n)
{
  b.tgeom = tgeom;
  b.order.resize(n);
  b.cen.resize(3 * (size_t)n);
  b.lo.resize(3 * (size_t)n);
  b.hi.resize(3 * (size_t)n);
  for (uint32_t k = 0; k < n; k++)
  {
    b.order[k] = k;
    b.tri_box(k);
  }
  b.build_root(n);
}
} // namespace

extern "C" int bvh_check(const double *tgeom, uint32_t n, uint32_t out[5], double *cost)
{
  if (n == 0)
    return 1;
  BvhBuild a, again;
  build(a, tgeom, n);
  build(again, tgeom, n);
  if (a.order != again.order || a.nodes != again.nodes || a.depth != again.depth)
    return 10; // not deterministic
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t t : a.order)
  {
    if (t >= n || seen[t])
      return 11; // not a permutation
    seen[t] = 1;
  }
  int least = 0;
  while (((uint64_t)PT_BVH_LEAF << least) < n)
    least++;
  if (a.max_depth != least || a.depth > std::max(least, 1))
    return 12;
  Walk w{a};
  double box[6];
  const uint32_t total = w.visit(0u, box, 0, false);
  if (w.err)
    return w.err;
  if (total != n || w.next_first != n || w.longest != a.depth)
    return 13;
  out[0] = (uint32_t)a.depth;
  out[1] = (uint32_t)a.max_depth;
  out[2] = (uint32_t)(a.nodes.size() / PT_BVH_SRC_DOUBLES);
  out[3] = w.leaves;
  out[4] = a.used_area_splits ? 1u : 0u;
  *cost = a.tree_cost();
  return 0;
}
