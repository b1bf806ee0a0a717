/* pt_trace.h -- one trace_path() call (raytracer.c:482-554: trace_step), one cast_ray() call (raytracer.c:556-641: whitted_step), and the
 * epilogue every body shares (per-pixel mean + gamma-5 tonemap, raytracer.c:212-220; the coalesced tile store; counters).
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_TRACE_H
#define PT_TRACE_H

/* ---- one trace_path() call (raytracer.c:482-554).  Returns true when the path ended; P.Ls
 * then holds the finished sample's radiance. -------------------------------------------- */
/* VARIANT: 0 literal scan / 1 filtered scan.  REFRACT: scene has M_REFRACTION materials.
 * CHECKER: scene has M_CHECKERED materials (atan2 / fmod code; its polynomial constants
 * would otherwise be hoisted into -- and spilled from -- registers of every scene).
 * TRIS: scene has triangles.  FILT_LDS: the filter table is staged in LDS (small scenes). */
/* One round of random_in_unit_sphere (raytracer.c:231-241): x, y, z drawn in that order.
 * Returns true when the round must be repeated.  Reference: while (sqrt(len2) > 1); with a
 * correctly rounded sqrt, sqrt(x) > 1  <=>  x > 1 + 2^-52 (x = 1 + 2^-52 still rounds to 1.0),
 * so the square root is taken once, after the last round (tests/test_host.py checks the
 * equivalence around the boundary). */
__device__ __forceinline__ bool rejection_round(uint64_t &rng, V3 &q, double &len2)
{
  q.x = rnd_pm1(rng);
  q.y = rnd_pm1(rng);
  q.z = rnd_pm1(rng);
  len2 = v_dot(q, q);
  return len2 > 1.0000000000000002;
}

/* random_on_hemisphere's tail (:242-253) and the cos_theta of :549 for an accepted sample q */
__device__ __forceinline__ V3 hemisphere_from_sample(const V3 &q, double len2, const V3 &n, double &weight)
{
  /* len2 is 0 or >= 2^-60 (coordinates are multiples of 2^-30): sqrt_unscaled's domain.
   * len2 == 0 needs three draws of exactly 2^30 (probability 2^-93); the reference aborts
   * there (assert in vec3_normalize, vector.h:56) */
  const double len = sqrt_unscaled(len2);
  V3 nd = v_scale(q, rcp_unscaled(len));
  /* :250-252 flip into the normal's hemisphere, :549 cos_theta = dot(flipped, n): negating a
   * vector negates its dot product exactly, so the second dot product is the first with the
   * sign the flip gave it */
  const double side = v_dot(nd, n);
  /* (the conditional negations as one sign mask XORed into the four high words: the same doubles -- negation is the
   * sign bit -- in five instructions instead of four negations and four selects) */
  const uint32_t flip = side < 0 ? 0x80000000u : 0u;
  auto signed_by = [flip](double x) { return __hiloint2double((int)((uint32_t)__double2hiint(x) ^ flip), __double2loint(x)); };
  nd = {signed_by(nd.x), signed_by(nd.y), signed_by(nd.z)};
  weight = signed_by(side);
  return nd;
}

/* The closest hit of one intersect() call, kept between the two halves of trace_step when
 * the pooled kernel postpones the walk of the triangle hierarchy. */
struct HitRec
{
  double min_t, bary_u, bary_v;
  int best;
  bool depth_ok; /* the call ran the scan at all (:487) */
  /* DEFER_DIR: a diffuse hit whose new direction is still to be sampled.  P.d holds the
   * NORMAL meanwhile, P.T lacks the factor albedo * cos; dir_slot / dir_scale say which
   * albedo (material slot, checker factor). */
  bool need_dir;
  uint32_t dir_slot; /* bit 31: the hit is on a hull facet whose stored normal points outward (PT_HULL_PLUS) */
  double dir_scale;
  /* the ray this call sends on starts on a hull facet and leaves on its outer side by more than the launch's
   * margin: it cannot meet a triangle (pt_build_hull_flags); kernels with parked walks skip the probe for it */
  bool leaving;
  TriLast last; /* kernels with M_CHECKERED code and triangles only */
};

/* MODE 0: the whole call.  MODE 1: the first half only -- depth test, the flat scan WITHOUT the
 * hierarchy walk; the result so far goes to *rec and nothing else changes.  MODE 2: the second
 * half only, from *rec (which the caller may have completed with bvh_traverse).
 * DEFER_DIR: a diffuse hit does not sample its direction here; the caller does (HitRec). */
template <int VARIANT, bool REFRACT, bool CHECKER, bool TRIS, bool FILT_LDS, int MODE = 0, bool DEFER_DIR = false,
          bool SPH_LDS = false, bool FILT_MEM = false, class STK = PendStack>
__device__ __forceinline__ bool trace_step(const SceneCtx &S, Path &P, uint32_t &n_casts,
                                           unsigned long long *diag_ptr, const STK &stack, int &stack_n,
                                           HitRec *rec = nullptr, const uint32_t *prim_pairs = nullptr)
{
  V3 add = {S.bg, S.bg, S.bg}; /* what this call contributes if the path ends here */
  bool path_ends = true;
  const V3 o = P.o, d = P.d;
  /* DEFER_DIR callers (the pooled and parked-walk kernels) flush a path's radiance to the pixel sums every trip: P.Ls is
   * zero on entry, and this call's one term -- the hit's emission whether the path goes on or dies in the roulette,
   * BACKGROUND if it found nothing or ran out of depth -- is the throughput AT ENTRY times `add`.  Formed once (below,
   * before the throughput changes) instead of accumulated into P.Ls in two places (three products, three "+ 0" the compiler may not fold, three more
   * additions and six selects per trip).  The static body (DEFER_DIR = false: M_REFRACTION beyond the pooled kernel, cast_ray) keeps the general form. */
  constexpr bool ONE_TERM = DEFER_DIR; /* (also with REFRACT -- the pooled refraction kernel: a refractive hit contributes its one emission term like any
                                        * other hit, the two children only divide the throughput between them) */
  HitRec local;
  HitRec &H = (MODE == 0 && !DEFER_DIR) ? local : *rec;

  if (MODE != 2)
  {
    H.depth_ok = P.depth <= S.max_depth;
    H.min_t = S.t_start; /* DBL_MAX */
    H.best = -1;
    H.bary_u = 0;
    H.bary_v = 0;
    H.last.idx = -1;
    H.last.u = 0;
    H.last.v = 0;
    if (H.depth_ok)
    {
      n_casts++;
      /* ---- intersect(): closest hit, strict <, index order (:393-464) ---- */
      if (VARIANT == 0)
      {
        /* the literal scan: spheres, then triangles, every lane on the same primitive */
        for (uint32_t i = 0; i < S.n_sph; i++)
          exact_sphere(S.geom + PT_GEOM_STRIDE * i, i, o, d, H.min_t, H.best);
        for (uint32_t i = 0; i < S.n_tri; i++)
          exact_triangle<false, CHECKER && TRIS>(S.tri + 9 * (size_t)i, S.n_sph + i, o, d, H.min_t, H.best, H.bary_u, H.bary_v,
                                                 &H.last);
      }
      else
        scan_filtered<TRIS, TRIS && !FILT_LDS, FILT_LDS, MODE == 0, CHECKER && TRIS, SPH_LDS, FILT_MEM>(
            S.geom, S.tri, (FILT_LDS || SPH_LDS) ? S.filt_lds : S.filt, S.near_R2, S.n_sph, S.n_sph + S.n_tri, o, d, H.min_t, H.best,
            H.bary_u, H.bary_v, diag_ptr, S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri, S.filt_shift, &H.last, S.stale_uv, S.tri32, prim_pairs, S.big,
            (TRIS && FILT_LDS && !(CHECKER && TRIS)) ? &S.mesh_bound : nullptr); /* (not where hit.u / hit.v follow EVERY passing triangle: same thing,
                                                                               * a triangle the ray passes lies inside the ball -- but keep that path as it was) */
    }
    if (MODE == 1)
      return false;
  }
  const double min_t = H.min_t;
  const int best = H.best;
  if (TRIS && (MODE != 0 || DEFER_DIR))
    H.leaving = false;
  /* ONE_TERM: BACKGROUND for every lane here, the hit's emission over it inside the hit's own branch below -- both before
   * anything touches the throughput, so no second copy of it has to live to the end of the call (nine 64-bit register
   * moves per trip), and written under the branches' lane masks, so no selects either */
  if (ONE_TERM)
    P.Ls = v_scale(P.T, S.bg);

  if (H.depth_ok)
  {
    if (best >= 0)
    {
      DIAG(8, 1);
      DIAG_LANES(9);
      /* ---- the winner's hit record (:406-411 / :428-431) ---- */
      V3 p = v_add(o, v_scale(d, min_t)); /* point_at :257 */
      /* (ONE_TERM kernels: the hit point becomes the path's origin here, for every lane that hit -- a path that ends below
       * never reads it again -- so that it is formed in the origin's own registers instead of copied there at the end) */
      if (ONE_TERM)
        P.o = p;
      V3 n;
      uint32_t slot, hull = 0u;
      double tex_u = 0, tex_v = 0;
      const bool is_tri = TRIS && (uint32_t)best >= S.n_sph;
      if (!is_tri)
      {
        const double *g = S.geom + PT_GEOM_STRIDE * best;
        /* |p - c| is r to within a few ulps of the coordinates, r in [1e-100, 1e100]
         * (rt_hip_scene_create): inside the domains of sqrt_unscaled and rcp_unscaled.  (Were
         * p to round onto c exactly, IEEE gives 0 * inf and this 0 * NaN: NaN both.) */
        const V3 pc = v_sub(p, ld3(g));
        n = v_scale(pc, rcp_unscaled(sqrt_unscaled(v_dot(pc, pc))));
        slot = (uint32_t)best;
      }
      else
      {
        const uint32_t ti = (uint32_t)best - S.n_sph;
        n = ld3(S.tri_normal + 3 * (size_t)ti);
        slot = S.tri_object[ti];
        hull = slot & (PT_HULL_PLUS | PT_HULL_MINUS);
        slot &= ~(PT_HULL_PLUS | PT_HULL_MINUS);
        /* the margin's error bound (hull_margin_for) assumes the ray that found this facet travelled no farther
         * than 2 near_R (its origin is then within 1.0001 x that + the facet's size of v0): a hit from farther away
         * -- a bounce off a far point of a wall-sized sphere -- lands less precisely on the plane.  (near_R^2 is
         * in SGPRs already; this kernel has none to spare for a constant of its own.) */
        if (TRIS && (MODE != 0 || DEFER_DIR) && !(min_t * min_t <= 4.0 * S.near_R2))
          hull = 0u;
      }
      const double *m = S.mat + PT_MAT_STRIDE * slot;
      const double prob = m[0];
      V3 albedo = ld3(m + 1);
      const V3 emission = ld3(m + 4);
      const uint32_t flags = (uint32_t)__double_as_longlong(m[7]);

      add = emission; /* a path that dies in the roulette returns emission (:502) */
      if (ONE_TERM)
        P.Ls = v_mul(P.T, emission);
      /* russian roulette :497-502: the draw is always consumed */
      if (rnd(P.rng) < prob)
      {
        path_ends = false;
        double checker_scale = 1.0;
        bool dir_deferred = false;
        if (CHECKER && (flags & PT_FLAG_CHECKER))
        {
          /* hit.u / hit.v as the scan leaves them: the LAST passing triangle's if the ray passes
           * any triangle (TriLast; a winning triangle passes, so it is covered), else the closest
           * sphere's (:410-411) */
          if (!(TRIS && H.last.idx >= 0))
          {
            tex_u = atan2_tab(n.x, n.z, S.atan_tab) / (2 * kPi) + 0.5; /* :410-411 */
            tex_v = n.y * 0.5 + 0.5;
          }
          else
          {
            /* :154-167 barycentric blend of the texture coordinates */
            const double *tx = S.tri_tex + 6 * (size_t)((uint32_t)H.last.idx - S.n_sph);
            const double lu = H.last.u, lv = H.last.v;
            double w0 = 1 - lu - lv;
            tex_u = (tx[0] * w0 + tx[2] * lu) + tx[4] * lv;
            tex_v = (tx[1] * w0 + tx[3] * lu) + tx[5] * lv;
          }
          /* checkered_texture :386-391, M = 100000 (:508) */
          double on = (double)((frac1(tex_u * 100000.0) > 0.5) ^ (frac1(tex_v * 100000.0) < 0.5)); /* fmod(., 1): frac1 */
          double c = 0.3 * (1 - on) + 0.7 * on;
          albedo = v_scale(albedo, c);
          checker_scale = c;
        }
        V3 nd;
        double weight = 1.0;
        bool split = false;
        if (REFRACT && (flags & PT_FLAG_REFRACT))
        {
          /* :514-529.  fresnel = mix(pow(1 - facing, 3), 1, 0.1); refract() with the
           * CLAMP_BETWEEN quirk (raytracer.h:30: cosi == 1 always) and iot = 1:
           *   eta = 1, k = 1 - eta*eta*(1 - cosi*cosi) = 1, n' = -N,
           *   refract(I) = I*eta + n'*(eta*cosi - sqrtf(k)) = I*1 + (-N)*0      (:354-373)
           * i.e. child A goes back along the incoming ray; child B is the mirror direction.
           * Both are normalised (:523, :526).  B waits on the stack with its share kr. */
          const double facing = -v_dot(d, n);
          const double fresnel = 1 * 0.1 + cube(1 - facing) * (1 - 0.1); /* pow(x, 3): a weight, see cube() */
          const double kr = fresnel, kt = (1 - fresnel) * 1.0;
          const V3 in = v_scale(d, -1);
          const V3 nn = v_scale(n, -1);
          const double coef = 1.0 * 1.0 - (double)sqrtf(1.0f);
          const V3 refr = v_add(v_scale(in, 1.0), v_scale(nn, coef));
          nd = v_normalize_fast(refr);
          const V3 refl = v_normalize_fast(v_sub(v_scale(d, 1), v_scale(n, 2 * v_dot(v_scale(d, 1), n))));
          const V3 base = v_mul(P.T, albedo);
          /* both children start on the facet and leave on the side the ray came from (A goes back along it, B is its mirror
           * image): on a hull facet neither can meet a triangle again (pt_build_hull_flags; the margin as for the mirror
           * and the diffuse bounce) -- kernels with parked walks skip the probe for them.  B's flag waits with it on the
           * stack, in bit 16 of its depth word. */
          int depth_b = P.depth + 1;
          if (TRIS && (MODE != 0 || DEFER_DIR))
          {
            const double an = v_dot(nd, n), bn = v_dot(refl, n);
            H.leaving = (hull & PT_HULL_PLUS) ? (an > S.hull_margin) : ((hull & PT_HULL_MINUS) ? (-an > S.hull_margin) : false);
            const bool b_leaves = (hull & PT_HULL_PLUS) ? (bn > S.hull_margin) : ((hull & PT_HULL_MINUS) ? (-bn > S.hull_margin) : false);
            depth_b |= b_leaves ? 0x10000 : 0;
          }
          if (stack_n < stack.capacity)
            stack.push(stack_n++, p, refl, v_scale(base, kr), depth_b);
          if (!ONE_TERM)
            P.Ls = v_add(P.Ls, v_mul(P.T, emission));
          P.T = v_scale(base, kt);
          split = true;
        }
        else if (flags & PT_FLAG_MIRROR)
        {
          /* reflect :349-352; direction left un-normalised (:542) */
          const double dn = v_dot(d, n);
          nd = v_sub(d, v_scale(n, 2 * dn));
          if (ONE_TERM)
            P.T = v_mul(P.T, albedo); /* here, under the branch's own lane mask, instead of three products and six selects below */
          /* nd . n = -(d . n) up to rounding far below the margin */
          if (TRIS && (MODE != 0 || DEFER_DIR))
            H.leaving = (hull & PT_HULL_PLUS) ? (-dn > S.hull_margin) : ((hull & PT_HULL_MINUS) ? (dn > S.hull_margin) : false);
        }
        else
        {
          /* random_on_hemisphere :231-253 */
          if (DEFER_DIR)
          {
            H.need_dir = true;
            H.dir_slot = slot | (hull & PT_HULL_PLUS); /* the bounce goes into the stored normal's hemisphere */
            H.dir_scale = checker_scale;
            dir_deferred = true;
            nd = n; /* P.d carries the normal until the caller has the sample */
          }
          else
          {
            V3 q;
            double len2;
            int tries = 0;
            bool again;
            do
            {
              DIAG(10, 1);
              DIAG_LANES(11);
              again = rejection_round(P.rng, q, len2);
            } while (again && ++tries < 100);
            nd = hemisphere_from_sample(q, len2, n, weight);
          }
        }
        /* L = e + albedo (.) (L_next * cos)  ==>  forward form */
        if (!split)
        {
          if (!ONE_TERM)
            P.Ls = v_add(P.Ls, v_mul(P.T, emission));
          if (!ONE_TERM && !(DEFER_DIR && dir_deferred))
            P.T = v_mul(P.T, (flags & PT_FLAG_MIRROR) ? albedo : v_scale(albedo, weight));
        }
        if (!ONE_TERM)
          P.o = p;
        P.d = nd;
        P.depth++;
      }
    }
  }
  if (path_ends)
  {
    if (!ONE_TERM)
      P.Ls = v_add(P.Ls, v_mul(P.T, add));
    if (REFRACT && stack_n > 0)
    {
      /* this branch of the tree is done: resume the most recent pending child; the RNG
       * stream simply continues, as it does across the reference's two recursive calls */
      stack.pop(--stack_n, P.o, P.d, P.T, P.depth);
      if (TRIS && (MODE != 0 || DEFER_DIR))
      { /* (the child's hull-facet flag: see the push) */
        H.leaving = (P.depth & 0x10000) != 0;
        P.depth &= 0xFFFF;
      }
      path_ends = false;
    }
  }
  return path_ends;
}

/* ---- one cast_ray() call (raytracer.c:556-641), the Whitted integrator on the other side of
 * render()'s `#if 1` (:207-211).  Same contract as trace_step: returns true when the sample is
 * finished.  One fixed point light (:567-568), Phong terms in the LIGHT's colour (1,1,1)
 * times the object colour (:586-603), a shadow ray with no distance limit (:570-572:
 * intersect(.., NULL) reports any hit in front of the point), a normalised mirror child
 * (:609-615) and a "refracted" child that, with the CLAMP_BETWEEN quirk, goes straight on
 * (:617-628).  Children are weighted by scalars, so the forward form carries a scalar weight
 * in P.T; a hit with both M_REFLECTION and M_REFRACTION traces the mirror child first and
 * parks the other on the pending-ray stack.  No random draws after the camera jitter. */
template <bool TRIS, bool FILT_LDS, bool STACK>
__device__ __forceinline__ bool whitted_step(const SceneCtx &S, Path &P, uint32_t &n_casts,
                                             unsigned long long *diag_ptr, const PendStack &stack, int &stack_n)
{
  V3 add = {S.bg, S.bg, S.bg}; /* depth limit or no hit: BACKGROUND (:561-564) */
  bool path_ends = true;
  const V3 o = P.o, d = P.d;

  if (P.depth <= S.max_depth)
  {
    n_casts++;
    double min_t = S.t_start;
    int best = -1;
    double bary_u = 0, bary_v = 0;
    TriLast last = {-1, 0, 0}; /* hit.u / hit.v of the scan: the last passing triangle's (TriLast) */
    scan_filtered<TRIS, TRIS && !FILT_LDS, FILT_LDS, true, TRIS>(S.geom, S.tri, FILT_LDS ? S.filt_lds : S.filt, S.near_R2,
                                                                 S.n_sph, S.n_sph + S.n_tri, o, d, min_t, best, bary_u,
                                                                 bary_v, diag_ptr, S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri,
                                                                 S.filt_shift, &last, S.stale_uv, S.tri32, nullptr, S.big);
    if (best >= 0)
    {
      const V3 p = v_add(o, v_scale(d, min_t));
      V3 n;
      uint32_t slot;
      const bool is_tri = (uint32_t)best >= S.n_sph;
      if (!is_tri)
      {
        const V3 pc = v_sub(p, ld3(S.geom + PT_GEOM_STRIDE * best));
        n = v_scale(pc, 1.0 / sqrt_unscaled(v_dot(pc, pc)));
        slot = (uint32_t)best;
      }
      else
      {
        const uint32_t ti = (uint32_t)best - S.n_sph;
        n = ld3(S.tri_normal + 3 * (size_t)ti);
        slot = S.tri_object[ti] & ~(PT_HULL_PLUS | PT_HULL_MINUS);
      }
      const uint32_t flags = (uint32_t)__double_as_longlong(S.mat[PT_MAT_STRIDE * slot + 7]);
      V3 color = ld3(S.color_raw + 3 * (size_t)slot);

      /* shadow ray :570-572 */
      const V3 light_pos = {2, 7, 2};
      const V3 ldir = v_normalize(v_sub(light_pos, p));
      n_casts++;
      double shadow_t = S.t_start, su = 0, sv = 0;
      int blocker = -1;
      scan_filtered<TRIS, TRIS && !FILT_LDS, FILT_LDS>(S.geom, S.tri, FILT_LDS ? S.filt_lds : S.filt, S.near_R2, S.n_sph,
                                                       S.n_sph + S.n_tri, p, ldir, shadow_t, blocker, su, sv, diag_ptr,
                                                       S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri, S.filt_shift, nullptr, false,
                                                       S.tri32, nullptr, S.big); /* (a shadow ray asks "any hit?": pruning walls that cannot be the
                                                                                  * CLOSEST hit never removes the closest one, so a hit stays a hit) */
      const double lit = blocker >= 0 ? 0.0 : 1.0;

      if (flags & PT_FLAG_CHECKER)
      {
        double tex_u, tex_v;
        if (!(TRIS && last.idx >= 0))
        {
          tex_u = atan2_tab(n.x, n.z, S.atan_tab) / (2 * kPi) + 0.5; /* :410-411 */
          tex_v = n.y * 0.5 + 0.5;
        }
        else
        {
          const double *tx = S.tri_tex + 6 * (size_t)((uint32_t)last.idx - S.n_sph);
          const double w0 = 1 - last.u - last.v;
          tex_u = (tx[0] * w0 + tx[2] * last.u) + tx[4] * last.v;
          tex_v = (tx[1] * w0 + tx[3] * last.u) + tx[5] * last.v;
        }
        /* checkered_texture :386-391 with M = 10 (:583) */
        const double on = (double)((frac1(tex_u * 10.0) > 0.5) ^ (frac1(tex_v * 10.0) < 0.5)); /* fmod(., 1): frac1 */
        color = v_scale(color, 0.3 * (1 - on) + 0.7 * on);
      }

      /* :586-603; light_color = (1,1,1), so each term is the same scalar in all channels */
      const double ka = 0.25, kd = 0.5, ks = 0.8, alpha = 10.0;
      const double n_dot_l = v_dot(n, ldir);
      const double diffuse = 1.0 * (kd * (0.0 > n_dot_l ? 0.0 : n_dot_l)); /* MAX(0.0, x) */
      const V3 reflected = v_sub(ldir, v_scale(n, 2 * v_dot(ldir, n)));
      const V3 view = v_normalize(v_sub(p, o));
      const double v_dot_r = v_dot(view, reflected);
      (void)alpha;
      const double specular = 1.0 * (ks * pow10(v_dot_r > 0.0 ? v_dot_r : 0.0)); /* pow(MAX(x, 0.0), alpha = 10): pow10() */
      const double shade = 1.0 * ka + (specular + diffuse) * lit;
      const V3 surface = v_scale(color, shade);
      add = surface;

      const bool mirror = (flags & PT_FLAG_MIRROR) != 0, glass = (flags & PT_FLAG_REFRACT) != 0;
      if (mirror || glass)
      {
        double kr = 1.0, kt = 0.0;
        V3 through = d;
        if (glass)
        {
          const double facing = -v_dot(d, n);
          const double fresnel = 1 * 0.1 + cube(1 - facing) * (1 - 0.1); /* mix() :255; pow(x, 3): cube() */
          kr = fresnel; /* :622 -- also the weight of an M_REFLECTION child of the same hit */
          kt = (1 - fresnel) * 0.5;
          /* refract(I, N, 1.0) :354-373 with cosi == 1: I*1 + (-N)*(1*1 - sqrtf(1)) */
          const double coef = 1.0 * 1.0 - (double)sqrtf(1.0f);
          through = v_normalize(v_add(v_scale(d, 1.0), v_scale(v_scale(n, -1), coef)));
        }
        P.Ls = v_add(P.Ls, v_mul(P.T, surface));
        const V3 weight = P.T;
        if (mirror)
        {
          const V3 refl = v_normalize(v_sub(d, v_scale(n, 2 * v_dot(d, n))));
          /* STACK = false: the launcher has checked that no material carries both flags */
          if (STACK && glass && stack_n < stack.capacity)
            stack.push(stack_n++, p, through, v_scale(weight, kt), P.depth + 1);
          P.d = refl;
          P.T = v_scale(weight, kr);
        }
        else
        {
          P.d = through;
          P.T = v_scale(weight, kt);
        }
        P.o = p;
        P.depth++;
        path_ends = false;
      }
    }
  }
  if (path_ends)
  {
    P.Ls = v_add(P.Ls, v_mul(P.T, add));
    if (STACK && stack_n > 0)
    {
      stack.pop(--stack_n, P.o, P.d, P.T, P.depth);
      path_ends = false;
    }
  }
  return path_ends;
}

/* ---- epilogue shared by both kernels: coalesced tile store + counters ------------------- */

/* per-pixel mean (raytracer.c:215) and gamma-5 tonemap (:218-220) of one tile from its
 * fixed-point sums; thread 3 t + c handles channel c of pixel t */
/* nan_mask[c]: bit t set = channel c of pixel t received a NaN sample.  The reference's fp64 sum
 * carries a NaN to the pixel (raytracer.c:212-215) and CLAMP(NaN) = 1 stores byte 255 (:218); an
 * integer sum cannot, so the pooled kernels flag such samples apart and the pixel becomes NaN
 * here.  (Samples are otherwise finite and within the scale's bound: emission is finite and the
 * throughput at most 1, rt_hip_render_tiles_chunked.) */
__device__ __forceinline__ void finish_pixels(const PtLaunch &L, const unsigned long long *sums,
                                              const unsigned long long *nan_mask, uint32_t tile, float *out_f,
                                              uint8_t *out_b)
{
  /* thread = (pixel, channel): 192 threads, one pow each (a pixel per thread kept three waves waiting on the first) */
  if (threadIdx.x < PT_TILE_PIXELS * 3)
  {
    const uint32_t t = threadIdx.x / 3u, c = threadIdx.x - 3u * t;
    const bool inside = (tile % L.tiles_x) * PT_TILE + (t & 7u) < (uint32_t)L.width &&
                        (tile / L.tiles_x) * PT_TILE + (t >> 3) < (uint32_t)L.height;
    const double inv_s = 1.0 / (double)L.samples;
    double mean = ((double)(long long)sums[threadIdx.x] * L.acc_inv_scale) * inv_s;
    const double quiet_nan = __longlong_as_double(0x7FF8000000000000ll);
    mean = ((nan_mask[c] >> t) & 1ull) ? quiet_nan : mean;
    out_f[threadIdx.x] = inside ? (float)mean : 0.f;
    out_b[threadIdx.x] = inside ? tonemap(mean) : 0;
  }
}

/* slot = index of the tile in the compact output; with_pixels = false when this workgroup
 * only contributed a sample chunk (pt_resolve_tiles writes the pixels then) */
__device__ __forceinline__ void store_tile(const PtLaunch &L, const float *out_f, const uint8_t *out_b,
                                           const unsigned long long *wg_stats, uint32_t tile, uint32_t slot,
                                           uint32_t n_prims, bool with_pixels, bool count_samples)
{
  /* 192 floats = 768 contiguous bytes per tile */
  if (with_pixels && threadIdx.x < PT_TILE_PIXELS * 3)
    L.tiles_rgb[(size_t)slot * (PT_TILE_PIXELS * 3) + threadIdx.x] = out_f[threadIdx.x];
  if (with_pixels && L.tiles_rgb8 && threadIdx.x < PT_TILE_PIXELS * 3 / 4)
    reinterpret_cast<uint32_t *>(L.tiles_rgb8)[(size_t)slot * (PT_TILE_PIXELS * 3 / 4) + threadIdx.x] =
        reinterpret_cast<const uint32_t *>(out_b)[threadIdx.x];
  if (L.stats && threadIdx.x == 0)
  {
    const unsigned long long rays = wg_stats[0], casts = wg_stats[1];
    atomicAdd(&L.stats[0], rays);
    atomicAdd(&L.stats[1], casts);
    atomicAdd(&L.stats[2], casts * (unsigned long long)n_prims);
  }
  if (L.stats && count_samples && threadIdx.x == 64)
  {
    const uint32_t tx0 = (tile % L.tiles_x) * PT_TILE, ty0 = (tile / L.tiles_x) * PT_TILE;
    const uint32_t cw = min((uint32_t)PT_TILE, (uint32_t)L.width - tx0);
    const uint32_t ch = min((uint32_t)PT_TILE, (uint32_t)L.height - ty0);
    atomicAdd(&L.stats[3], (unsigned long long)cw * ch * (unsigned long long)L.samples);
  }
}

#endif /* PT_TRACE_H */
