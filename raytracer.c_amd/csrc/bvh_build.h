/* bvh_build.h -- the host-side builder of the triangle hierarchy (rt_hip_scene_create, rt_hip_shim.hip).
 * Plain C++ (no HIP): also compiled by g++ into the CPU test of its invariants (tests/bvh_harness.cpp, tests/test_host.py). */
#ifndef BVH_BUILD_H
#define BVH_BUILD_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pt_device.h"

/* ---- bounding-volume hierarchy over triangles ----
 * Binary tree, leaves of <= PT_BVH_LEAF triangles, of the LEAST depth such leaves allow (max_depth = the smallest D with
 * PT_BVH_LEAF * 2^D >= n: a lane's traversal stack in LDS is sized by the depth, and one level more costs config 5's kernel its
 * fourth workgroup per CU).  Within that depth a node is split by the surface-area heuristic over 64 bins of the centroid
 * bounds on each axis (round 4; cost = area(left box) * count(left) + area(right box) * count(right), only splits that leave
 * both sides a subtree of the remaining depth: count <= PT_BVH_LEAF * 2^(levels left)); where no such split exists -- or with
 * RT_HIP_BVH_MEDIAN=1, round 1-3's builder, for A/B -- by the median on the longest axis of the centroid bounds, which always
 * does.  On config 5's 10,240-triangle sphere: node visits per walked ray 15.3 -> 13.9, leaf pre-tests 21.0 -> 18.3, leaves
 * 2.1 -> 1.6 at the same depth of 10.  A node stores the boxes of its two
 * children, so the device tests both with one set of packed-fp32 instructions, descends into
 * the nearer one first and keeps the other on a small per-lane stack (bvh_traverse).  The
 * hierarchy only decides WHICH triangles get the exact test; the exact test and the
 * (t, index) tie rule decide the result, so any valid hierarchy gives the linear scan's answer. */
struct BvhBuild
{
  const double *tgeom;              /* n_tri x 9: v0, e1, e2 */
  std::vector<uint32_t> order;      /* triangle indices, permuted in place */
  std::vector<double> nodes;        /* PT_BVH_SRC_DOUBLES per node: the two children's boxes + refs */
  std::vector<double> cen, lo, hi;  /* per triangle: centroid, box */
  std::vector<uint8_t> bin_of;      /* per triangle: its bin on the axis being tried (scratch of the split search) */
  int depth = 0;                    /* inner nodes on the longest root-to-leaf path */
  int max_depth = 0;                /* the least depth leaves of PT_BVH_LEAF allow: no split may exceed it */
  bool median_only = false;         /* the tree being built splits at the median everywhere */
  bool used_area_splits = false;    /* build_root kept the surface-area tree (it was the cheaper one by tree_cost) */

  static double half_area(const double *l, const double *h)
  {
    const double dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
    return dx * dy + dy * dz + dz * dx;
  }

  /* The surface-area split of order[begin, end) on the bins of the centroid bounds [cl, ch]: partitions the range (stably, so the
   * build is deterministic) and returns the size of its left part, or 0 when no admissible split exists.  `cap` = the most
   * triangles a side may hold and still fit a subtree of the remaining depth. */
  uint32_t sah_split(uint32_t begin, uint32_t end, const double *cl, const double *ch, uint64_t cap)
  {
    constexpr int NB = 64;
    const uint32_t n = end - begin;
    double best = 1e300;
    int best_axis = -1, best_split = 0;
    uint32_t best_left = 0;
    for (int axis = 0; axis < 3; axis++)
    {
      const double ext = ch[axis] - cl[axis];
      if (!(ext > 0) || !(ext < 1e300))
        continue;
      const double scale = NB / ext;
      uint32_t cnt[NB] = {0};
      double blo[NB][3], bhi[NB][3];
      for (int b = 0; b < NB; b++)
        for (int k = 0; k < 3; k++)
        {
          blo[b][k] = 1e300;
          bhi[b][k] = -1e300;
        }
      for (uint32_t i = begin; i < end; i++)
      {
        const uint32_t t = order[i];
        const int b = std::min(NB - 1, std::max(0, (int)((cen[3 * t + axis] - cl[axis]) * scale)));
        cnt[b]++;
        for (int k = 0; k < 3; k++)
        {
          blo[b][k] = std::fmin(blo[b][k], lo[3 * t + k]);
          bhi[b][k] = std::fmax(bhi[b][k], hi[3 * t + k]);
        }
      }
      /* right-hand boxes and counts by a sweep from the top, then the left-hand ones with the candidate splits */
      double r_area[NB];
      uint32_t r_cnt[NB];
      {
        double l3[3] = {1e300, 1e300, 1e300}, h3[3] = {-1e300, -1e300, -1e300};
        uint32_t c = 0;
        for (int b = NB - 1; b >= 1; b--)
        {
          for (int k = 0; k < 3; k++)
          {
            l3[k] = std::fmin(l3[k], blo[b][k]);
            h3[k] = std::fmax(h3[k], bhi[b][k]);
          }
          c += cnt[b];
          r_cnt[b] = c;
          r_area[b] = c ? half_area(l3, h3) : 0.0;
        }
      }
      double l3[3] = {1e300, 1e300, 1e300}, h3[3] = {-1e300, -1e300, -1e300};
      uint32_t c = 0;
      for (int sp = 1; sp < NB; sp++) /* bins [0, sp) go left */
      {
        for (int k = 0; k < 3; k++)
        {
          l3[k] = std::fmin(l3[k], blo[sp - 1][k]);
          h3[k] = std::fmax(h3[k], bhi[sp - 1][k]);
        }
        c += cnt[sp - 1];
        if (c == 0 || c == n || c > cap || r_cnt[sp] > cap)
          continue;
        const double cost = half_area(l3, h3) * c + r_area[sp] * r_cnt[sp];
        if (cost < best)
        {
          best = cost;
          best_axis = axis;
          best_split = sp;
          best_left = c;
        }
      }
    }
    if (best_axis < 0)
      return 0;
    const double scale = NB / (ch[best_axis] - cl[best_axis]);
    for (uint32_t i = begin; i < end; i++)
    {
      const uint32_t t = order[i];
      bin_of[t] = (uint8_t)std::min(NB - 1, std::max(0, (int)((cen[3 * t + best_axis] - cl[best_axis]) * scale)));
    }
    std::stable_partition(order.begin() + begin, order.begin() + end, [&](uint32_t t) { return bin_of[t] < best_split; });
    return best_left;
  }

  void tri_box(uint32_t t)
  {
    const double *g = tgeom + 9 * (size_t)t;
    for (int k = 0; k < 3; k++)
    {
      const double a = g[k], b = g[k] + g[3 + k], c = g[k] + g[6 + k]; /* v0, v0+e1, v0+e2 */
      /* v1 = v0 + e1 is re-rounded here; the slack is absorbed by the device-side margins,
       * which are orders of magnitude larger than one ulp of a coordinate */
      lo[3 * t + k] = std::fmin(a, std::fmin(b, c));
      hi[3 * t + k] = std::fmax(a, std::fmax(b, c));
      cen[3 * t + k] = (a + b + c) / 3.0;
    }
  }

  /* Builds the subtree over order[begin, end) and returns its reference (a leaf reference or
   * the index of its inner node); box[0..5] receives its bounds. */
  uint32_t build(uint32_t begin, uint32_t end, double *box, int level)
  {
    double cl[3] = {1e300, 1e300, 1e300}, ch[3] = {-1e300, -1e300, -1e300};
    for (int k = 0; k < 3; k++)
    {
      box[k] = 1e300;
      box[3 + k] = -1e300;
    }
    for (uint32_t i = begin; i < end; i++)
      for (int k = 0; k < 3; k++)
      {
        const uint32_t t = order[i];
        box[k] = std::fmin(box[k], lo[3 * t + k]);
        box[3 + k] = std::fmax(box[3 + k], hi[3 * t + k]);
        cl[k] = std::fmin(cl[k], cen[3 * t + k]);
        ch[k] = std::fmax(ch[k], cen[3 * t + k]);
      }
    if (end - begin <= PT_BVH_LEAF)
      return PT_BVH_LEAF_FLAG | (begin << PT_BVH_COUNT_BITS) | (end - begin);
    const uint32_t me = (uint32_t)(nodes.size() / PT_BVH_SRC_DOUBLES);
    nodes.resize(nodes.size() + PT_BVH_SRC_DOUBLES, 0.0);
    depth = std::max(depth, level + 1);
    /* either side must fit a subtree of the levels left below this node (the median always does: n <= 2 cap here) */
    const int left_levels = max_depth - level - 1;
    const uint64_t cap = left_levels >= 0 && left_levels < 40 ? ((uint64_t)PT_BVH_LEAF << left_levels) : (uint64_t)PT_BVH_LEAF;
    uint32_t mid = median_only ? 0u : sah_split(begin, end, cl, ch, cap);
    if (mid != 0u)
      mid += begin;
    else
    {
      int axis = 0;
      if (ch[1] - cl[1] > ch[axis] - cl[axis]) axis = 1;
      if (ch[2] - cl[2] > ch[axis] - cl[axis]) axis = 2;
      mid = begin + (end - begin) / 2;
      std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                       [&](uint32_t a, uint32_t b) { return cen[3 * a + axis] < cen[3 * b + axis]; });
    }
    double b0[6], b1[6];
    const uint32_t refs[2] = {build(begin, mid, b0, level + 1), build(mid, end, b1, level + 1)};
    double *n = &nodes[PT_BVH_SRC_DOUBLES * (size_t)me]; /* after the recursion: nodes may have moved */
    memcpy(n, b0, sizeof b0);
    memcpy(n + 6, b1, sizeof b1);
    memcpy(n + 12, refs, sizeof refs);
    return me;
  }

  /* What a walk of this tree costs on average, per ray that meets the root's box, in the walk's own instruction counts: a
   * visited inner node tests its two children's boxes (~60 VALU instructions, bvh_test_children), a visited leaf puts its
   * triangles through the fp32 pre-test (~45 each); a box is visited in proportion to its surface area (the usual
   * assumption behind the heuristic).  tools/bvh_sim.py and the device's PT_DIAG counts agree with its ranking on config 5. */
  double tree_cost() const
  {
    if (nodes.empty())
      return 0.0;
    double root[6];
    const double *n0 = &nodes[0];
    for (int k = 0; k < 3; k++)
    {
      root[k] = std::fmin(n0[k], n0[6 + k]);
      root[3 + k] = std::fmax(n0[3 + k], n0[9 + k]);
    }
    const double a_root = half_area(root, root + 3);
    if (!(a_root > 0) || !(a_root < 1e300))
      return 0.0;
    double cost = 60.0; /* the root's own visit */
    for (size_t i = 0; i < nodes.size() / PT_BVH_SRC_DOUBLES; i++)
    {
      const double *n = &nodes[i * PT_BVH_SRC_DOUBLES];
      uint32_t refs[2];
      memcpy(refs, n + 12, sizeof refs);
      for (int c = 0; c < 2; c++)
      {
        const uint32_t count = refs[c] & ((1u << PT_BVH_COUNT_BITS) - 1u);
        if ((refs[c] & PT_BVH_LEAF_FLAG) && count == 0)
          continue; /* the empty second child of a one-leaf mesh */
        const double p = half_area(n + 6 * c, n + 6 * c + 3) / a_root;
        cost += p * ((refs[c] & PT_BVH_LEAF_FLAG) ? 45.0 * count : 60.0);
      }
    }
    return cost;
  }

  void build_tree(uint32_t n_tri)
  {
    double box[6];
    nodes.clear();
    depth = 0;
    const uint32_t ref = build(0, n_tri, box, 0);
    if (ref & PT_BVH_LEAF_FLAG)
    { /* a mesh that fits one leaf still gets a root node: child 0 = the leaf, child 1 = an empty leaf */
      nodes.assign(PT_BVH_SRC_DOUBLES, 0.0);
      const uint32_t refs[2] = {ref, PT_BVH_LEAF_FLAG};
      memcpy(&nodes[0], box, sizeof box);
      memcpy(&nodes[6], box, sizeof box);
      memcpy(&nodes[12], refs, sizeof refs);
      depth = 1;
    }
  }

  /* The whole mesh (order = the caller's initial order, normally the identity).  Both trees are built -- the area splits are
   * greedy, and under the depth cap a lopsided split near the root can use up the slack (PT_BVH_LEAF * 2^max_depth - n) that
   * the levels below would have needed: on a mesh with one dense cluster and little slack the capped area tree came out 20 %
   * dearer than the median tree -- and the one with the lower tree_cost() is kept, so the area heuristic can only help. */
  void build_root(uint32_t n_tri)
  {
    max_depth = 0;
    while (((uint64_t)PT_BVH_LEAF << max_depth) < n_tri)
      max_depth++;
#ifdef PT_DEV_KERNELS
    const char *e = getenv("RT_HIP_BVH_MEDIAN"); /* development builds (A/B, and the two-builders-one-frame test): round 1-3's median builder alone */
    const bool median_asked = e && e[0] == '1';
#else
    const bool median_asked = false;
#endif
    bin_of.assign(n_tri, 0);
    const std::vector<uint32_t> order0 = order;
    median_only = true;
    build_tree(n_tri);
    used_area_splits = false;
    if (median_asked)
      return;
    const double cost_median = tree_cost();
    std::vector<uint32_t> order_m = order;
    std::vector<double> nodes_m = nodes;
    const int depth_m = depth;
    order = order0;
    median_only = false;
    build_tree(n_tri);
    if (tree_cost() < cost_median)
    {
      used_area_splits = true;
      return;
    }
    order.swap(order_m);
    nodes.swap(nodes_m);
    depth = depth_m;
  }
};

#endif /* BVH_BUILD_H */
