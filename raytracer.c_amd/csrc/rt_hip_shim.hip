/* rt_hip_shim.hip -- the C-ABI of include/rt_hip.h over the kernels of
 * pt_kernel.hip.  Thin by design: argument checks, the one-time conversion of
 * the reference's scene structs (Object raytracer.h:104-111, Vertex :61) into
 * the kernel's HBM layout (pt_device.h), launches, and the single-process
 * multi-GPU gather.  No CPU rendering path exists here: every failure is
 * reported, never papered over.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <vector>

#include "pt_device.h"
#include "rt_hip.h"
#include "bvh_build.h"

static_assert(sizeof(RtHipSphere) == 88, "RtHipSphere must match the reference Object");
static_assert(sizeof(RtHipVertex) == 40, "RtHipVertex must match the reference Vertex");
static_assert(sizeof(RtHipCamera) == 96, "RtHipCamera must match the reference Camera");
static_assert(RT_HIP_TILE == PT_TILE && RT_HIP_TILE_PIXELS == PT_TILE_PIXELS, "tile shape");
/* exact_triangle<UNSCALED> divides through rcp_unscaled, valid for 2^-500 <= |a| <= 2^500 with |a| <= |e1||e2||d|: a launch
 * refuses near_R >= RT_NEAR_R_LIMIT, vertices lie within near_R / 1.5, so |e1|, |e2| < 2 near_R / 1.5 and |d| <= 1.0001 */
#define RT_NEAR_R_LIMIT 1e15
static_assert((2.0 * RT_NEAR_R_LIMIT / 1.5) * (2.0 * RT_NEAR_R_LIMIT / 1.5) * 1.0001 < 3.2e150 /* < 2^500 */,
              "the near_R limit keeps |e1||e2||d| inside rcp_unscaled's range");

/* The camera-dependent tables of a scene (packed-fp32 filter table, fp32 hierarchy nodes; both
 * depend on near_R, i.e. on the camera's distance) for one near_R.  Built once on the stream of
 * the first launch that needs them and immutable afterwards, so launches of one scene with
 * different cameras on different streams or threads never write a table another launch reads;
 * later launches on other streams wait on `built`.  A scene keeps up to RT_TABLE_SETS of them;
 * beyond that the least recently used one is recycled after every stream that read it (`readers`) is done. */
struct TableSet
{
  double near_R = 0;
  float *filt = nullptr, *bvh_nodes = nullptr;
  hipEvent_t built = nullptr;
  /* one "last read" event PER READER STREAM: a single event re-recorded by whichever launch releases last would
   * forget the readers on other streams (an event holds only its latest record), and the set could be recycled
   * under a kernel that still reads it (round-2 advisor finding) */
  std::vector<std::pair<hipStream_t, hipEvent_t>> readers;
  uint64_t stamp = 0;
  int users = 0;      /* launches between acquire_tables() and release_tables(): not recyclable */
  bool owned = false; /* allocated apart from the scene blob */
};
#define RT_TABLE_SETS 8

struct RtHipScene
{
  int device = 0;
  PtSceneView view{}; /* view.filt / view.bvh_nodes: storage of table set 0, inside the blob */
  mutable std::mutex table_mutex;
  mutable std::vector<TableSet> tables;
  mutable uint64_t table_clock = 0;
  size_t filt_bytes = 0, bvh_nodes_bytes = 0;
  /* workspace of the parked-walk kernels (scenes with a triangle hierarchy): in-use flags, then the rings
   * (pt_device.h).  ONE per device, shared by every scene on it and counted (park_acquire_ws / park_drop_ws): the 385 MB (round 4: 462; round 3: 406)
   * used to be allocated per scene -- a test suite's every 48 x 32 fuzz scene paid it, and scenes alive at the same
   * time each held a copy (round-3 advisor finding).  Sharing is safe between concurrent launches of different scenes:
   * a workgroup takes a slot with an atomic flag, and the pool has more slots per XCD than workgroups can be resident. */
  mutable char *park_ws = nullptr;
  mutable uint32_t park_slots_per_xcd = 0;
  mutable bool park_tried = false;
  void *blob = nullptr; /* one device allocation holding every array */
  double reach = 0;     /* >= |p| for every point p on a primitive of ordinary size (radius < 1000) */
  double max_emission = 0; /* max |emission component| over all materials */
  bool any_mirror_glass = false; /* a material with M_REFLECTION and M_REFRACTION: cast_ray traces two children */
  double max_center = 0; /* max |centre| over the spheres (rounded up) */
  /* the scene's LEADING wall-sized spheres (radius >= 1000), whole pairs of them, at most 2 PT_BIG_PAIRS: radius and
   * |centre| -- what big_prune_for needs to bound their hit-distance estimates (pt_filter.h, BigPrune) */
  int n_big = 0;
  double big_r[8] = {0}, big_c[8] = {0};
  /* bounding sphere of every triangle (bvh_probe): centre, radius, |centre| -- radius < 0: no triangles */
  double mesh_c[3] = {0, 0, 0}, mesh_R = -1, mesh_c_norm = 0;
  bool hull_flags = false; /* tri_object carries PT_HULL_PLUS / PT_HULL_MINUS (pt_build_hull_flags ran) */
};

namespace
{

thread_local char g_err[512] = "";
const volatile int *g_cancel = nullptr; /* rt_hip_set_cancel_flag */

int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do                                                                                                \
  {                                                                                                 \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(e_ == hipErrorOutOfMemory ? RT_HIP_ENOMEM : RT_HIP_ERUNTIME, "%s: %s", #expr,     \
                  hipGetErrorString(e_));                                                           \
  } while (0)

#define NCCL_TRY(expr)                                                                              \
  do                                                                                                \
  {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess)                                                                          \
      return fail(RT_HIP_ERUNTIME, "%s: %s", #expr, ncclGetErrorString(r_));                        \
  } while (0)

/* selects a device for the current scope and puts the previous one back */
struct DeviceScope
{
  int prev = -1;
  hipError_t status;
  explicit DeviceScope(int device)
  {
    status = hipGetDevice(&prev);
    if (status == hipSuccess && prev != device)
      status = hipSetDevice(device);
  }
  ~DeviceScope()
  {
    if (prev >= 0)
      (void)hipSetDevice(prev);
  }
};

int usable_devices()
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
  {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

/* host mirrors of the reference's vector.h operations (order matters) */
struct H3
{
  double x, y, z;
};
H3 h_sub(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
double h_dot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
H3 h_cross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
H3 h_scale(H3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
H3 h3(const double *p) { return {p[0], p[1], p[2]}; }

double max_abs3(const double *v) { return std::fmax(std::fabs(v[0]), std::fmax(std::fabs(v[1]), std::fabs(v[2]))); }

/* The pooled kernels' fixed-point sums rest on throughput <= 1, i.e. on albedo / MAX(albedo)
 * in [0, 1]: colours must be finite and non-negative; emission finite. */
bool material_ok(const double *color, const double *emission)
{
  for (int k = 0; k < 3; k++)
    if (!(color[k] >= 0.0) || !(color[k] <= 1e100) || !(std::fabs(emission[k]) <= 1e100))
      return false;
  return true;
}

void put_material(double *m, uint32_t flags, const double *color, const double *emission)
{
  /* raytracer.c:497: prob = MAX(albedo.x, MAX(albedo.y, albedo.z)) */
  double yz = color[1] > color[2] ? color[1] : color[2];
  double prob = color[0] > yz ? color[0] : yz;
  double inv = 1 / prob; /* :500 vec3_scalar_mult(albedo, 1 / prob) */
  m[0] = prob;
  m[1] = color[0] * inv;
  m[2] = color[1] * inv;
  m[3] = color[2] * inv;
  m[4] = emission[0];
  m[5] = emission[1];
  m[6] = emission[2];
  uint64_t bits = flags;
  memcpy(&m[7], &bits, sizeof bits);
}

/* One sphere's record of entry_src (pt_device.h): cx cy cz, r*r, |c| rounded up, 0. */
void sphere_entry(const double *center, double radius, double *g)
{
  g[0] = center[0];
  g[1] = center[1];
  g[2] = center[2];
  g[3] = radius * radius; /* raytracer.c:87 */
  /* |c|, rounded up: feeds the conservative phase-1 thresholds only, never a result */
  g[4] = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) * (1.0 + 1e-12);
  g[5] = 0.0; /* a sphere is rejected when its centre is behind the origin at all (raytracer.c:84) */
}

/* One triangle: tri_geom record (v0, e1, e2) and its entry_src record, the phase-1 bound: a
 * sphere around the centroid through the farthest vertex, slightly enlarged; feeds the
 * conservative filter only, never a result. */
void triangle_entry(const double *p0, const double *p1, const double *p2, double *g, double *b)
{
  H3 v0 = h3(p0), v1 = h3(p1), v2 = h3(p2);
  H3 e1 = h_sub(v1, v0), e2 = h_sub(v2, v0); /* raytracer.c:132-133 */
  g[0] = v0.x; g[1] = v0.y; g[2] = v0.z;
  g[3] = e1.x; g[4] = e1.y; g[5] = e1.z;
  g[6] = e2.x; g[7] = e2.y; g[8] = e2.z;
  H3 cen = {(v0.x + v1.x + v2.x) / 3.0, (v0.y + v1.y + v2.y) / 3.0, (v0.z + v1.z + v2.z) / 3.0};
  double rb2 = 0;
  const H3 vs[3] = {v0, v1, v2};
  for (int j = 0; j < 3; j++)
  {
    H3 dv = h_sub(vs[j], cen);
    rb2 = std::fmax(rb2, h_dot(dv, dv));
  }
  const double rb = std::sqrt(rb2) * (1.0 + 1e-9) + 1e-300;
  b[0] = cen.x;
  b[1] = cen.y;
  b[2] = cen.z;
  b[3] = rb * rb;
  b[4] = std::sqrt(h_dot(cen, cen)) * (1.0 + 1e-12);
  b[5] = rb;
}

/* ---- bounding-volume hierarchy over triangles: BvhBuild, in bvh_build.h (host C++, shared with the CPU test of its invariants) ---- */

/* Fault injection for tests (rt_hip_selftest_fail_alloc): which of the shim's optional device allocations behave as if
 * hipMalloc had failed, so that the fallback kernels are reachable -- and testable -- on a 288 GB device. */
std::atomic<uint32_t> g_fail_alloc{0};

/* the kernel the calling thread's last rt_hip_render_tiles* launched (rt_hip_last_launch_kernel) */
thread_local int g_last_kernel = -1;

/* The table set of `scene` for `near_R`, ready to be read by work submitted to `stream` after this
 * call (see TableSet).  *slot identifies it for release_tables(). */
int acquire_tables(const RtHipScene *scene, double near_R, hipStream_t stream, float **filt, float **bvh_nodes, size_t *slot)
{
  std::lock_guard<std::mutex> lock(scene->table_mutex);
  std::vector<TableSet> &tables = scene->tables;
  for (size_t k = 0; k < tables.size(); k++)
    if (tables[k].near_R == near_R)
    {
      HIP_TRY(hipStreamWaitEvent(stream, tables[k].built, 0));
      tables[k].stamp = ++scene->table_clock;
      tables[k].users++;
      *filt = tables[k].filt;
      *bvh_nodes = tables[k].bvh_nodes;
      *slot = k;
      return RT_HIP_OK;
    }
  size_t k = tables.size();
  if (k < RT_TABLE_SETS)
  {
    TableSet t;
    if (k == 0)
    { /* the storage inside the scene blob */
      t.filt = scene->view.filt;
      t.bvh_nodes = scene->view.bvh_nodes;
    }
    else
    {
      t.owned = true;
      HIP_TRY(hipMalloc(&t.filt, scene->filt_bytes ? scene->filt_bytes : 256));
      hipError_t e = hipMalloc(&t.bvh_nodes, scene->bvh_nodes_bytes ? scene->bvh_nodes_bytes : 256);
      if (e != hipSuccess)
      {
        (void)hipFree(t.filt);
        return fail(RT_HIP_ENOMEM, "hipMalloc of a hierarchy table: %s", hipGetErrorString(e));
      }
    }
    hipError_t e = hipEventCreateWithFlags(&t.built, hipEventDisableTiming);
    if (e != hipSuccess)
    {
      if (t.owned)
      {
        (void)hipFree(t.filt);
        (void)hipFree(t.bvh_nodes);
      }
      return fail(RT_HIP_ERUNTIME, "hipEventCreate: %s", hipGetErrorString(e));
    }
    tables.push_back(t);
  }
  else
  {
    /* recycle the least recently used set once every launch that reads it has finished */
    k = tables.size();
    for (size_t j = 0; j < tables.size(); j++)
      if (tables[j].users == 0 && (k == tables.size() || tables[j].stamp < tables[k].stamp))
        k = j;
    if (k == tables.size())
      return fail(RT_HIP_ELIMIT, "more than %d launches of one scene with different camera distances are being submitted at once",
                  RT_TABLE_SETS);
    HIP_TRY(hipEventSynchronize(tables[k].built));
    for (auto &r : tables[k].readers) /* every stream that ever read this set, not only the last one to release it */
      HIP_TRY(hipEventSynchronize(r.second));
  }
  TableSet &t = tables[k];
  t.near_R = near_R;
  t.stamp = ++scene->table_clock;
  {
    hipError_t e = pt_launch_build_tables(scene->view, near_R, t.filt, t.bvh_nodes, stream);
    if (e == hipSuccess) e = hipEventRecord(t.built, stream);
    if (e != hipSuccess)
    {
      t.near_R = -1.0; /* never matches a launch: the set is rebuilt (or recycled) by the next one */
      return fail(RT_HIP_ERUNTIME, "building the filter / hierarchy tables: %s", hipGetErrorString(e));
    }
  }
  t.users++;
  *filt = t.filt;
  *bvh_nodes = t.bvh_nodes;
  *slot = k;
  return RT_HIP_OK;
}

/* after the render that reads table set `slot` has been submitted to `stream` */
void release_tables(const RtHipScene *scene, size_t slot, hipStream_t stream)
{
  std::unique_lock<std::mutex> lock(scene->table_mutex);
  if (slot >= scene->tables.size())
    return;
  hipEvent_t ev = nullptr;
  {
    TableSet &t = scene->tables[slot];
    for (auto &r : t.readers)
      if (r.first == stream)
        ev = r.second;
    if (!ev && t.readers.size() >= 32)
    { /* a caller cycling through many streams: take the oldest reader's event out of the set, wait it out WITHOUT
       * the lock (other threads' launches of this scene go on meanwhile; the set cannot be recycled under the
       * waited-for reader because this launch still counts in `users`), and hand the event to this stream */
      hipEvent_t old = t.readers.front().second;
      t.readers.erase(t.readers.begin());
      lock.unlock();
      (void)hipEventSynchronize(old);
      lock.lock();
      ev = old;
      scene->tables[slot].readers.emplace_back(stream, ev);
    }
    else if (!ev)
    {
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess)
        t.readers.emplace_back(stream, ev);
      else
      { /* no event to remember this reader by: let it finish before the set can be recycled */
        ev = nullptr;
        (void)hipGetLastError();
        lock.unlock();
        (void)hipStreamSynchronize(stream);
        lock.lock();
      }
    }
  }
  TableSet &t = scene->tables[slot]; /* (the vector may have grown while the lock was away) */
  if (ev)
    (void)hipEventRecord(ev, stream);
  t.users--;
}

/* the per-device workspace of the parked-walk kernels (RtHipScene::park_ws) */
struct ParkPool
{
  char *ws = nullptr;
  int users = 0;
  uint32_t slots_per_xcd = 0;
};
std::mutex g_park_mutex;
ParkPool g_park[64];

size_t park_flag_bytes(uint32_t slots_per_xcd)
{
  return ((size_t)PT_PARK_XCDS * slots_per_xcd * sizeof(uint32_t) + 255) & ~(size_t)255;
}

/* -> the device's workspace (allocated and its flags zeroed at the first call), or nullptr when the allocation fails:
 * the scene then renders on the lane-waiting kernels, and rt_hip_kernel_name says so.  The current device is `device`. */
char *park_acquire_ws(int device, uint32_t *slots_per_xcd)
{
  if (device < 0 || device >= 64)
    return nullptr;
  if (g_fail_alloc.load() & RT_HIP_FAIL_ALLOC_PARK_WS) /* tests: behave as if the allocation had failed */
    return nullptr;
  std::lock_guard<std::mutex> lock(g_park_mutex);
  ParkPool &p = g_park[device];
  if (!p.ws)
  {
    const uint32_t per = pt_pool_slots_per_xcd(true);
    const size_t n_slots = (size_t)PT_PARK_XCDS * per;
    char *ws = nullptr;
    if (hipMalloc(&ws, park_flag_bytes(per) + n_slots * (PT_BLOCK / 64) * (size_t)PT_PARK_WAVE_BYTES) != hipSuccess)
    {
      (void)hipGetLastError();
      return nullptr;
    }
    /* the flags must be zero before a kernel on ANY stream looks at them (kernels leave them zero) */
    if (hipMemset(ws, 0, park_flag_bytes(per)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
    {
      (void)hipGetLastError();
      (void)hipFree(ws);
      return nullptr;
    }
    p.ws = ws;
    p.slots_per_xcd = per;
  }
  p.users++;
  *slots_per_xcd = p.slots_per_xcd;
  return p.ws;
}

/* a scene that held the device's workspace goes away (its launches have been waited for): the last one frees it */
void park_drop_ws(int device)
{
  if (device < 0 || device >= 64)
    return;
  std::lock_guard<std::mutex> lock(g_park_mutex);
  ParkPool &p = g_park[device];
  if (p.users > 0 && --p.users == 0)
  {
    (void)hipFree(p.ws);
    p.ws = nullptr;
    p.slots_per_xcd = 0;
  }
}

/* the per-device pool of pending-ray stacks of the two-child kernels (pt_scene_ctx.h, PendStack): flags, then
 * PT_PARK_XCDS x slots_per_xcd (pt_pool_slots_per_xcd: 160 on an MI355X) slots of `entries` x 10 fields x PT_PEND_COLUMNS doubles.  Sized by the deepest launch
 * seen so far (max_depth + 2 entries: 367 MB at the reference's MAX_DEPTH 5, 1.8 GB at the limit of 32); grown -- after
 * the device has drained -- when a launch needs more, never shrunk; rt_hip_release_cache() frees it. */
struct PendPool
{
  char *ws = nullptr;
  uint32_t entries = 0, columns = 0, slots_per_xcd = 0;
  uint32_t small_streak = 0; /* consecutive launches that needed at most a quarter of it (pend_pool_for: when it shrinks) */
};
/* a pool above PEND_SHRINK_ABOVE bytes is given up for one that fits after PEND_SHRINK_AFTER launches in a row needed at most a
 * quarter of it: one max_depth-29 glass-mesh launch used to pin 6.5 GB until rt_hip_release_cache() (round-4 advisor finding) */
constexpr size_t PEND_SHRINK_ABOVE = (size_t)1 << 30;
constexpr uint32_t PEND_SHRINK_AFTER = 16;
std::mutex g_pend_mutex; /* held from the pool lookup until the launch that uses it is enqueued (so that a growing
                          * launch's hipDeviceSynchronize covers every kernel that holds the old pointer) */
PendPool g_pend[64];

size_t pend_flag_bytes(uint32_t slots_per_xcd) { return ((size_t)PT_PARK_XCDS * slots_per_xcd * sizeof(uint32_t) + 255) & ~(size_t)255; }

/* caller holds g_pend_mutex; the current device is `device` */
int pend_pool_for(int device, uint32_t entries, uint32_t columns, PtLaunch &L)
{
  if (device < 0 || device >= 64)
    return fail(RT_HIP_ENODEV, "device %d: no pending-ray pool", device);
  PendPool &p = g_pend[device];
  if (p.ws && p.entries >= entries && p.columns >= columns)
  { /* large enough.  Far too large, for a while now?  Then the device drains once and the pool is rebuilt to fit. */
    const size_t have_bytes = (size_t)PT_PARK_XCDS * p.slots_per_xcd * p.entries * PT_PEND_FIELDS_HOST * p.columns * sizeof(double);
    if (have_bytes > PEND_SHRINK_ABOVE && (uint64_t)entries * columns * 4u <= (uint64_t)p.entries * p.columns)
    {
      if (++p.small_streak >= PEND_SHRINK_AFTER)
      {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(p.ws);
        p.ws = nullptr;
        p.entries = p.columns = p.small_streak = 0;
      }
    }
    else
      p.small_streak = 0;
  }
  if (p.entries < entries || p.columns < columns)
  {
    const bool asks_wide = columns > PT_PEND_COLUMNS && p.columns < columns; /* this launch is what asks for 4 x 512 stacks per slot */
    entries = std::max(entries, p.entries); /* grown in either direction, never shrunk */
    columns = std::max(columns, p.columns);
    const uint32_t per = p.slots_per_xcd ? p.slots_per_xcd : pt_pool_slots_per_xcd(false);
    const size_t slot_bytes = (size_t)entries * PT_PEND_FIELDS_HOST * columns * sizeof(double);
    const size_t n_slots = (size_t)PT_PARK_XCDS * per;
    /* tests: the request for 4 x 512 stacks per slot behaves as if hipMalloc had failed (BEFORE the old pool is given up) */
    if (asks_wide && (g_fail_alloc.load() & RT_HIP_FAIL_ALLOC_WIDE_PEND))
      return fail(RT_HIP_ENOMEM, "pending-ray pool for max_depth %u (%zu MB): allocation failure injected", entries - 2u,
                  (pend_flag_bytes(per) + n_slots * slot_bytes) >> 20);
    if (p.ws)
    {
      HIP_TRY(hipDeviceSynchronize());
      (void)hipFree(p.ws);
      p.ws = nullptr;
      p.entries = p.columns = 0;
    }
    char *ws = nullptr;
    hipError_t e = hipMalloc(&ws, pend_flag_bytes(per) + n_slots * slot_bytes);
    if (e != hipSuccess)
    {
      (void)hipGetLastError(); /* the caller may go on with a narrower pool: a later hipGetLastError() must not see this failure (round-4 advisor finding) */
      return fail(RT_HIP_ENOMEM, "pending-ray pool for max_depth %u (%zu MB): %s", entries - 2u,
                  (pend_flag_bytes(per) + n_slots * slot_bytes) >> 20, hipGetErrorString(e));
    }
    e = hipMemset(ws, 0, pend_flag_bytes(per));
    if (e == hipSuccess)
      e = hipStreamSynchronize(nullptr); /* the flags are zero before a kernel on any stream looks at them */
    if (e != hipSuccess)
    {
      (void)hipGetLastError();
      (void)hipFree(ws);
      return fail(RT_HIP_ERUNTIME, "pending-ray pool: %s", hipGetErrorString(e));
    }
    p.ws = ws;
    p.entries = entries;
    p.columns = columns;
    p.slots_per_xcd = per;
    p.small_streak = 0;
  }
  L.pend_flags = reinterpret_cast<uint32_t *>(p.ws);
  L.pend_ws = reinterpret_cast<double *>(p.ws + pend_flag_bytes(p.slots_per_xcd));
  L.pend_slots_per_xcd = p.slots_per_xcd;
  L.pend_entries = p.entries; /* slots are laid out for the pool's depth; a shallower launch uses a prefix of each */
  L.pend_slot_doubles = (uint64_t)p.entries * PT_PEND_FIELDS_HOST * p.columns;
  return RT_HIP_OK;
}

/* the per-device status word of render launches (PtLaunch.status, rt_hip_launch_status) */
std::mutex g_status_mutex;
uint32_t *g_status[64] = {nullptr};

/* the current device is `device` */
int status_word_for(int device, uint32_t **out)
{
  if (device < 0 || device >= 64)
    return fail(RT_HIP_ENODEV, "device %d: no status word", device);
  std::lock_guard<std::mutex> lock(g_status_mutex);
  if (!g_status[device])
  {
    uint32_t *w = nullptr;
    HIP_TRY(hipMalloc(&w, 256));
    hipError_t e = hipMemset(w, 0, 256);
    if (e == hipSuccess)
      e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
    {
      (void)hipFree(w);
      return fail(RT_HIP_ERUNTIME, "status word: %s", hipGetErrorString(e));
    }
    g_status[device] = w;
  }
  *out = g_status[device];
  return RT_HIP_OK;
}

/* reads and clears the device's status word; the current device is `device` */
int status_take(int device, uint32_t *flags)
{
  *flags = 0;
  uint32_t *w = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_status_mutex);
    w = (device >= 0 && device < 64) ? g_status[device] : nullptr;
  }
  if (!w)
    return RT_HIP_OK; /* nothing has been launched on this device */
  HIP_TRY(hipMemcpy(flags, w, sizeof *flags, hipMemcpyDeviceToHost));
  if (*flags)
    HIP_TRY(hipMemset(w, 0, sizeof *flags));
  return RT_HIP_OK;
}

int status_to_error(uint32_t flags)
{
  if (!flags)
    return RT_HIP_OK;
  return fail(RT_HIP_ERUNTIME, "render launch failed on the device:%s%s -- the affected tiles were not rendered (they read NaN / 255)",
              (flags & RT_HIP_FAIL_PEND_SLOT) ? " a workgroup found no free slot in the pending-ray pool;" : "",
              (flags & RT_HIP_FAIL_PARK_SLOT) ? " a workgroup found no free slot in the parked-walk workspace;" : "");
}

void pend_pools_release()
{
  std::lock_guard<std::mutex> lock(g_pend_mutex);
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (int d = 0; d < 64; d++)
    if (g_pend[d].ws)
    {
      (void)hipSetDevice(d);
      (void)hipDeviceSynchronize();
      (void)hipFree(g_pend[d].ws);
      g_pend[d] = PendPool();
    }
  {
    std::lock_guard<std::mutex> slock(g_status_mutex);
    for (int d = 0; d < 64; d++)
      if (g_status[d])
      {
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        (void)hipFree(g_status[d]);
        g_status[d] = nullptr;
      }
  }
  (void)hipSetDevice(prev);
}

uint32_t tiles_x_of(int width) { return ((uint32_t)width + PT_TILE - 1) / PT_TILE; }
uint32_t tiles_y_of(int height) { return ((uint32_t)height + PT_TILE - 1) / PT_TILE; }

int check_params(const RtHipParams *p)
{
  if (!p)
    return fail(RT_HIP_EINVAL, "params is NULL");
  if (p->width < 2 || p->height < 2)
    return fail(RT_HIP_EINVAL, "width and height must be >= 2 (the reference divides by width-1, height-1)");
  if (p->samples < 1 || p->samples > (1 << 26))
    return fail(RT_HIP_EINVAL, "samples must be in [1, 2^26]");
  if (p->max_depth < 0 || p->max_depth > 1000000)
    return fail(RT_HIP_EINVAL, "max_depth out of range");
  if (p->integrator != RT_HIP_TRACE_PATH && p->integrator != RT_HIP_CAST_RAY)
    return fail(RT_HIP_EINVAL, "integrator must be RT_HIP_TRACE_PATH (0) or RT_HIP_CAST_RAY (1)");
  if ((uint64_t)p->width * (uint64_t)p->height > 0xFFFFFFFFull)
    return fail(RT_HIP_EINVAL, "image has more than 2^32 pixels");
  if (p->width > (1 << 20) || p->height > (1 << 20))
    return fail(RT_HIP_EINVAL, "width and height must not exceed 2^20 (the kernel's exact-quotient shortcut)");
  return RT_HIP_OK;
}

} // namespace

extern "C" {

const char *rt_hip_last_error(void) { return g_err; }

void rt_hip_set_cancel_flag(const volatile int *flag) { g_cancel = flag; }

int rt_hip_device_count(void) { return usable_devices(); }

int rt_hip_device_info(int device, char *name, size_t name_cap, int *compute_units)
{
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (name && name_cap)
    snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units)
    *compute_units = prop.multiProcessorCount;
  return RT_HIP_OK;
}

} /* extern "C" */

namespace
{
int scene_create_impl(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes, size_t n_meshes,
                      int device, RtHipScene **out_scene);
int render_image_impl(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes, size_t n_meshes,
                      const RtHipCamera *camera, const RtHipParams *params, int n_devices, float *h_image_rgb,
                      uint8_t *h_image_rgb8, uint64_t *h_stats, double *kernel_seconds);
void release_cache_impl();
uint64_t cache_builds_impl();
int set_device_map_impl(const int *map, int n);
void last_phases_impl(double out[3]);
} // namespace

extern "C" {

/* C++ exceptions (std::bad_alloc from the staging vectors) must not cross the C boundary */
int rt_hip_scene_create(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes,
                        size_t n_meshes, int device, RtHipScene **out_scene)
{
  try
  {
    return scene_create_impl(spheres, n_spheres, meshes, n_meshes, device, out_scene);
  }
  catch (const std::bad_alloc &)
  {
    return fail(RT_HIP_ENOMEM, "host allocation failed while staging the scene");
  }
  catch (...)
  {
    return fail(RT_HIP_ERUNTIME, "unexpected C++ exception in rt_hip_scene_create");
  }
}

void rt_hip_release_cache(void) { release_cache_impl(); }

void rt_hip_last_image_phases(double seconds[3])
{
  if (seconds)
    last_phases_impl(seconds);
}

int rt_hip_set_device_map(const int *map, int n)
{
  try
  {
    return set_device_map_impl(map, n);
  }
  catch (...)
  {
    return fail(RT_HIP_ENOMEM, "host allocation failed in rt_hip_set_device_map");
  }
}

uint64_t rt_hip_cache_builds(void) { return cache_builds_impl(); }

int rt_hip_render_image(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes,
                        size_t n_meshes, const RtHipCamera *camera, const RtHipParams *params,
                        int n_devices, float *h_image_rgb, uint8_t *h_image_rgb8, uint64_t *h_stats,
                        double *kernel_seconds)
{
  try
  {
    return render_image_impl(spheres, n_spheres, meshes, n_meshes, camera, params, n_devices, h_image_rgb,
                             h_image_rgb8, h_stats, kernel_seconds);
  }
  catch (const std::bad_alloc &)
  {
    return fail(RT_HIP_ENOMEM, "host allocation failed in rt_hip_render_image");
  }
  catch (...)
  {
    return fail(RT_HIP_ERUNTIME, "unexpected C++ exception in rt_hip_render_image");
  }
}

} /* extern "C" */

namespace
{

int scene_create_impl(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes, size_t n_meshes,
                      int device, RtHipScene **out_scene)
{
  if (!out_scene)
    return fail(RT_HIP_EINVAL, "out_scene is NULL");
  *out_scene = nullptr;
  if ((n_spheres && !spheres) || (n_meshes && !meshes))
    return fail(RT_HIP_EINVAL, "NULL scene array with non-zero count");
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d (found %d)", device, usable_devices());

  size_t n_tri = 0;
  bool any_checker = false, any_refract = false, any_mirror_glass = false;
  const uint32_t both = PT_FLAG_MIRROR | PT_FLAG_REFRACT;
  for (size_t i = 0; i < n_spheres; i++)
  {
    any_refract |= (spheres[i].flags & PT_FLAG_REFRACT) != 0;
    any_mirror_glass |= (spheres[i].flags & both) == both;
    any_checker |= (spheres[i].flags & PT_FLAG_CHECKER) != 0;
    if (!material_ok(spheres[i].color, spheres[i].emission))
      return fail(RT_HIP_EINVAL, "sphere %zu: colour must be finite and >= 0, emission finite", i);
    if (!(std::fabs(spheres[i].radius) >= 1e-100) || !(std::fabs(spheres[i].radius) <= 1e100))
      return fail(RT_HIP_ELIMIT, "sphere %zu: |radius| %g outside [1e-100, 1e100]", i, spheres[i].radius);
  }
  for (size_t m = 0; m < n_meshes; m++)
  {
    any_refract |= (meshes[m].flags & PT_FLAG_REFRACT) != 0;
    any_mirror_glass |= (meshes[m].flags & both) == both;
    if (!material_ok(meshes[m].color, meshes[m].emission))
      return fail(RT_HIP_EINVAL, "mesh %zu: colour must be finite and >= 0, emission finite", m);
    if (meshes[m].num_triangles && !meshes[m].vertices)
      return fail(RT_HIP_EINVAL, "mesh %zu has triangles but no vertices", m);
    any_checker |= (meshes[m].flags & PT_FLAG_CHECKER) != 0;
    n_tri += meshes[m].num_triangles;
  }
  if (n_spheres + n_meshes > 0xFFFFFFu || n_tri > 0x7FFFFFFFu - n_spheres)
    return fail(RT_HIP_ELIMIT, "scene too large");
  const size_t n_mat = n_spheres + n_meshes;

  /* ---- build the kernel layout on the host (pt_device.h) ---- */
  double reach = 0, max_emission = 0, max_center = 0;
  bool wide_range = false;
  for (size_t i = 0; i < n_spheres; i++)
    max_emission = std::fmax(max_emission, max_abs3(spheres[i].emission));
  for (size_t m = 0; m < n_meshes; m++)
    max_emission = std::fmax(max_emission, max_abs3(meshes[m].emission));
  std::vector<double> geom(PT_ENTRY_SRC_STRIDE * (n_spheres + n_tri)), mat(PT_MAT_STRIDE * n_mat), tgeom(9 * n_tri), tnorm(3 * n_tri),
      ttex(6 * n_tri);
  std::vector<uint32_t> tobj(n_tri);
  std::vector<double> craw(3 * n_mat), geom4(PT_GEOM_STRIDE * n_spheres);
  for (size_t i = 0; i < n_spheres; i++)
    memcpy(&craw[3 * i], spheres[i].color, 3 * sizeof(double));
  for (size_t m = 0; m < n_meshes; m++)
    memcpy(&craw[3 * (n_spheres + m)], meshes[m].color, 3 * sizeof(double));
  for (size_t i = 0; i < n_spheres; i++)
  {
    double *g = &geom[PT_ENTRY_SRC_STRIDE * i];
    sphere_entry(spheres[i].center, spheres[i].radius, g);
    memcpy(&geom4[PT_GEOM_STRIDE * i], g, PT_GEOM_STRIDE * sizeof(double));
    max_center = std::fmax(max_center, g[4]);
    wide_range |= !(g[4] <= 1e17) || !(std::fabs(spheres[i].radius) <= 1e17);
    if (std::fabs(spheres[i].radius) < 1000.0) /* wall-sized spheres would only loosen the filter */
      reach = std::fmax(reach, g[4] + std::fabs(spheres[i].radius));
    put_material(&mat[PT_MAT_STRIDE * i], spheres[i].flags, spheres[i].color, spheres[i].emission);
  }
  size_t t = 0;
  for (size_t m = 0; m < n_meshes; m++)
    for (size_t k = 0; k < 3 * meshes[m].num_triangles; k++)
    {
      const double *q = meshes[m].vertices[k].pos;
      const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
      /* a non-finite coordinate is refused here: fmax(reach, NaN) would drop it silently, the hierarchy builder would sort
       * NaN centroids (no strict weak ordering) and quantise them (undefined), and the kernels' UNSCALED division would lose
       * its range argument (round-4 advisor finding).  The reference has no such check -- and no meaning for such a triangle:
       * every comparison of its test (raytracer.c:137-150) is false or the triangle is never hit. */
      if (!(len <= 1e300))
        return fail(RT_HIP_EINVAL, "mesh %zu: vertex %zu has a non-finite coordinate", m, k);
      /* (a vertex far out needs no flag of its own: it is part of `reach`, and a launch refuses
       * near_R = 1.5 (|camera| + reach) + 1 >= 1e15 as "not a usable finite bound" -- so every vertex a kernel ever sees
       * lies within 6.7e14 of the origin, which is what exact_triangle's UNSCALED division rests on, see below) */
      reach = std::fmax(reach, len);
    }
  for (size_t m = 0; m < n_meshes; m++)
  {
    put_material(&mat[PT_MAT_STRIDE * (n_spheres + m)], meshes[m].flags, meshes[m].color, meshes[m].emission);
    for (size_t k = 0; k < meshes[m].num_triangles; k++, t++)
    {
      const RtHipVertex *v = meshes[m].vertices + 3 * k;
      double *g = &tgeom[9 * t];
      triangle_entry(v[0].pos, v[1].pos, v[2].pos, g, &geom[PT_ENTRY_SRC_STRIDE * (n_spheres + t)]);
      /* calculate_surface_normal :42-45: normalize(cross(v2-v0, v1-v0)) */
      H3 c = h_cross(h3(g + 6), h3(g + 3));
      H3 n = h_scale(c, 1.0 / std::sqrt(h_dot(c, c)));
      tnorm[3 * t + 0] = n.x; tnorm[3 * t + 1] = n.y; tnorm[3 * t + 2] = n.z;
      for (int j = 0; j < 3; j++)
      {
        ttex[6 * t + 2 * j + 0] = v[j].tex[0];
        ttex[6 * t + 2 * j + 1] = v[j].tex[1];
      }
      tobj[t] = (uint32_t)(n_spheres + m);
    }
  }

  /* ---- bounding sphere of all triangles, as the kernels see them (v0, v0 + e1, v0 + e2): centre = middle of
   * their bounds, radius = the farthest corner, with slack for the roundings of e1, e2 and of the exact test's
   * own barycentric limits ---- */
  double mesh_c[3] = {0, 0, 0}, mesh_R = -1;
  bool mesh_round = false; /* the sphere's silhouette is no larger than the mean silhouette of the triangles' box */
  if (n_tri != 0)
  {
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    auto corner = [&](size_t tri, int k, double *p) {
      const double *g = &tgeom[9 * tri];
      for (int a = 0; a < 3; a++)
        p[a] = k == 0 ? g[a] : g[a] + g[3 * k + a];
    };
    for (size_t i = 0; i < n_tri; i++)
      for (int k = 0; k < 3; k++)
      {
        double p[3];
        corner(i, k, p);
        for (int a = 0; a < 3; a++)
        {
          lo[a] = std::fmin(lo[a], p[a]);
          hi[a] = std::fmax(hi[a], p[a]);
        }
      }
    for (int a = 0; a < 3; a++)
      mesh_c[a] = 0.5 * lo[a] + 0.5 * hi[a];
    double r2 = 0;
    for (size_t i = 0; i < n_tri; i++)
      for (int k = 0; k < 3; k++)
      {
        double p[3];
        corner(i, k, p);
        const double dx = p[0] - mesh_c[0], dy = p[1] - mesh_c[1], dz = p[2] - mesh_c[2];
        r2 = std::fmax(r2, dx * dx + dy * dy + dz * dz);
      }
    mesh_R = std::sqrt(r2) * (1.0 + 1e-9) + 1e-300;
    if (!(mesh_R < HUGE_VAL)) /* non-finite input: a bound that keeps every ray */
      mesh_R = HUGE_VAL;
    const double a = hi[0] - lo[0], b = hi[1] - lo[1], c = hi[2] - lo[2];
    mesh_round = 3.14159265358979 * mesh_R * mesh_R <= 0.5 * (a * b + b * c + c * a); /* false for NaN / inf */
  }

  /* ---- hierarchy over the triangles ---- */
  BvhBuild bvh;
  /* built for every scene with triangles: the small-scene kernels scan them through the flat
   * filter instead, but the general in-memory kernels (too-large scenes, cast_ray with
   * two-child materials) always walk the hierarchy */
  if (n_tri > 0)
  {
    bvh.tgeom = tgeom.data();
    bvh.order.resize(n_tri);
    bvh.cen.resize(3 * n_tri);
    bvh.lo.resize(3 * n_tri);
    bvh.hi.resize(3 * n_tri);
    for (uint32_t k = 0; k < (uint32_t)n_tri; k++)
    {
      bvh.order[k] = k;
      bvh.tri_box(k);
    }
    if (n_tri >= (1u << (31 - PT_BVH_COUNT_BITS)))
      return fail(RT_HIP_ELIMIT, "%zu triangles exceed the hierarchy's leaf references (2^%d)", n_tri, 31 - PT_BVH_COUNT_BITS);
    bvh.build_root((uint32_t)n_tri);
    if (bvh.depth > PT_BVH_STACK)
      return fail(RT_HIP_ELIMIT, "triangle hierarchy depth %d exceeds the traversal stack (%d)", bvh.depth, PT_BVH_STACK);
  }
  const size_t n_bvh_nodes = bvh.nodes.size() / PT_BVH_SRC_DOUBLES;
  std::vector<double> tgeom_leaf(9 * bvh.order.size());
  for (size_t k = 0; k < bvh.order.size(); k++)
    memcpy(&tgeom_leaf[9 * k], &tgeom[9 * (size_t)bvh.order[k]], 9 * sizeof(double));

  /* ---- one device blob ---- */
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t off_geom = 0;
  const size_t off_mat = off_geom + pad(geom.size() * 8);
  const size_t off_craw = off_mat + pad(mat.size() * 8);
  const size_t off_geom4 = off_craw + pad(craw.size() * 8);
  const size_t off_tgeom = off_geom4 + pad(geom4.size() * 8);
  const size_t off_tnorm = off_tgeom + pad(tgeom.size() * 8);
  const size_t off_ttex = off_tnorm + pad(tnorm.size() * 8);
  const size_t off_tobj = off_ttex + pad(ttex.size() * 8);
  const size_t off_filt = off_tobj + pad(tobj.size() * 4);
  /* + 1 pair: the scan's software pipeline reads one pair past the end */
  const size_t filt_bytes = pt_filt_bytes((uint32_t)n_spheres, (uint32_t)n_tri);
  const size_t off_bvh_src = off_filt + pad(filt_bytes);
  const size_t off_bvh_nodes = off_bvh_src + pad(bvh.nodes.size() * 8);
  const size_t bvh_nodes_bytes = n_bvh_nodes * PT_BVH_NODE_WORDS * 4;
  const size_t off_bvh_tri = off_bvh_nodes + pad(bvh_nodes_bytes);
  const size_t off_tgeom_leaf = off_bvh_tri + pad(bvh.order.size() * 4);
  const size_t total = off_tgeom_leaf + pad(tgeom_leaf.size() * 8) + 256;

  DeviceScope scope(device);
  HIP_TRY(scope.status);
  RtHipScene *sc = new (std::nothrow) RtHipScene;
  if (!sc)
    return fail(RT_HIP_ENOMEM, "host allocation failed");
  sc->device = device;
  hipError_t e = hipMalloc(&sc->blob, total);
  if (e != hipSuccess)
  {
    delete sc;
    return fail(RT_HIP_ENOMEM, "hipMalloc(%zu): %s", total, hipGetErrorString(e));
  }
  char *base = static_cast<char *>(sc->blob);
  auto up = [&](size_t off, const void *src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpy(base + off, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  e = up(off_geom, geom.data(), geom.size() * 8);
  if (e == hipSuccess) e = up(off_mat, mat.data(), mat.size() * 8);
  if (e == hipSuccess) e = up(off_craw, craw.data(), craw.size() * 8);
  if (e == hipSuccess) e = up(off_geom4, geom4.data(), geom4.size() * 8);
  if (e == hipSuccess) e = up(off_tgeom, tgeom.data(), tgeom.size() * 8);
  if (e == hipSuccess) e = up(off_tnorm, tnorm.data(), tnorm.size() * 8);
  if (e == hipSuccess) e = up(off_ttex, ttex.data(), ttex.size() * 8);
  if (e == hipSuccess) e = up(off_tobj, tobj.data(), tobj.size() * 4);
  if (e == hipSuccess) e = up(off_bvh_src, bvh.nodes.data(), bvh.nodes.size() * 8);
  if (e == hipSuccess) e = up(off_bvh_tri, bvh.order.data(), bvh.order.size() * 4);
  if (e == hipSuccess) e = up(off_tgeom_leaf, tgeom_leaf.data(), tgeom_leaf.size() * 8);
  if (e != hipSuccess)
  {
    (void)hipFree(sc->blob);
    delete sc;
    return fail(RT_HIP_ERUNTIME, "scene upload: %s", hipGetErrorString(e));
  }
  sc->view.entry_src = reinterpret_cast<const double *>(base + off_geom);
  sc->view.filt = reinterpret_cast<float *>(base + off_filt);
  sc->view.bvh_src = reinterpret_cast<const double *>(base + off_bvh_src);
  sc->view.bvh_nodes = reinterpret_cast<float *>(base + off_bvh_nodes);
  sc->view.bvh_tri = reinterpret_cast<const uint32_t *>(base + off_bvh_tri);
  sc->view.tri_geom_leaf = reinterpret_cast<const double *>(base + off_tgeom_leaf);
  sc->view.n_bvh_nodes = (uint32_t)n_bvh_nodes;
  sc->view.bvh_depth = (uint32_t)bvh.depth;
  for (int a = 0; a < 3; a++)
    sc->mesh_c[a] = mesh_c[a];
  sc->mesh_R = mesh_R;
  sc->mesh_c_norm = std::sqrt(mesh_c[0] * mesh_c[0] + mesh_c[1] * mesh_c[1] + mesh_c[2] * mesh_c[2]) * (1.0 + 1e-12);
  sc->filt_bytes = filt_bytes;
  sc->bvh_nodes_bytes = bvh_nodes_bytes;
  sc->view.material = reinterpret_cast<const double *>(base + off_mat);
  sc->view.color_raw = reinterpret_cast<const double *>(base + off_craw);
  sc->view.geom4 = reinterpret_cast<const double *>(base + off_geom4);
  sc->view.tri_geom = reinterpret_cast<const double *>(base + off_tgeom);
  sc->view.tri_normal = reinterpret_cast<const double *>(base + off_tnorm);
  sc->view.tri_tex = reinterpret_cast<const double *>(base + off_ttex);
  sc->view.tri_object = reinterpret_cast<const uint32_t *>(base + off_tobj);
  if (n_tri != 0 && n_tri <= PT_HULL_MAX_TRIS && mesh_R >= 0 && mesh_R < 1e150)
  {
    /* hull facets (pt_build_hull_flags): tau = 2^-43 x the triangles' extent -- 64 u sigma_max extent with u = 2^-53
     * and the shape limit sigma_max = 16, well above the residual of a facet's own corners and of coplanar
     * neighbours (~u sigma extent) */
    const double extent = sc->mesh_c_norm + mesh_R;
    hipError_t he = pt_launch_build_hull_flags(sc->view.tri_geom, sc->view.tri_normal, (uint32_t)n_tri,
                                               1.1368683772161603e-13 * extent, reinterpret_cast<uint32_t *>(base + off_tobj), nullptr);
    if (he == hipSuccess)
      he = hipDeviceSynchronize();
    if (he != hipSuccess)
    {
      (void)hipFree(sc->blob);
      delete sc;
      return fail(RT_HIP_ERUNTIME, "hull flags: %s", hipGetErrorString(he));
    }
    sc->hull_flags = true;
  }
  sc->view.n_spheres = (uint32_t)n_spheres;
  sc->view.n_meshes = (uint32_t)n_meshes;
  sc->view.n_triangles = (uint32_t)n_tri;
  sc->view.any_checker = any_checker ? 1u : 0u;
  sc->view.mesh_round = mesh_round ? 1u : 0u;
  sc->view.any_refract = any_refract ? 1u : 0u;
#ifdef PT_DEV_KERNELS
  {
    /* development knob (tools/many_spheres.py): RT_HIP_FORCE_BIG=1 sends a small scene to the scalar-table _big kernels,
     * which it otherwise reaches only through a centre or radius beyond 1e17 */
    const char *fb = getenv("RT_HIP_FORCE_BIG");
    if (fb && fb[0] == '1')
      wide_range = true;
  }
#endif
  sc->view.wide_range = wide_range ? 1u : 0u;
  sc->max_center = max_center;
  {
    int nb = 0;
    while (nb < 8 && (size_t)nb < n_spheres && std::fabs(spheres[nb].radius) >= 1000.0 && std::fabs(spheres[nb].radius) <= 1e17)
    {
      sc->big_r[nb] = std::fabs(spheres[nb].radius);
      sc->big_c[nb] = geom[PT_ENTRY_SRC_STRIDE * (size_t)nb + 4];
      nb++;
    }
    sc->n_big = nb & ~1; /* whole pairs */
  }
  sc->reach = reach;
  sc->max_emission = max_emission;
  sc->any_mirror_glass = any_mirror_glass;
  sc->view.any_mirror_glass = any_mirror_glass ? 1u : 0u;
  *out_scene = sc;
  return RT_HIP_OK;
}

} // namespace

extern "C" {

void rt_hip_scene_destroy(RtHipScene *scene)
{
  if (!scene)
    return;
  {
    DeviceScope scope(scene->device);
    for (TableSet &t : scene->tables)
    {
      if (t.built) (void)hipEventSynchronize(t.built);
      for (auto &r : t.readers)
      {
        (void)hipEventSynchronize(r.second);
        (void)hipEventDestroy(r.second);
      }
      if (t.built) (void)hipEventDestroy(t.built);
      if (t.owned)
      {
        (void)hipFree(t.filt);
        (void)hipFree(t.bvh_nodes);
      }
    }
    if (scene->park_ws)
    { /* every launch of this scene must be past its last ring access before the pool can go */
      (void)hipDeviceSynchronize();
      park_drop_ws(scene->device);
    }
    (void)hipFree(scene->blob);
  }
  delete scene;
}

int rt_hip_scene_device(const RtHipScene *scene) { return scene ? scene->device : -1; }

size_t rt_hip_scene_primitives(const RtHipScene *scene)
{
  return scene ? (size_t)scene->view.n_spheres + scene->view.n_triangles : 0;
}

int rt_hip_scene_hull_facets(const RtHipScene *scene, uint32_t *n_plus, uint32_t *n_minus)
{
  if (!scene || !n_plus || !n_minus)
    return fail(RT_HIP_EINVAL, "rt_hip_scene_hull_facets: null argument");
  *n_plus = *n_minus = 0;
  const size_t n = scene->view.n_triangles;
  if (n == 0 || !scene->hull_flags)
    return 0;
  std::vector<uint32_t> obj(n);
  DeviceScope on(scene->device);
  HIP_TRY(on.status);
  HIP_TRY(hipMemcpy(obj.data(), scene->view.tri_object, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++)
  {
    *n_plus += (obj[i] & PT_HULL_PLUS) ? 1u : 0u;
    *n_minus += (obj[i] & PT_HULL_MINUS) ? 1u : 0u;
  }
  return 0;
}

const char *rt_hip_kernel_name(const RtHipScene *scene, uint32_t integrator)
{
  if (!scene)
    return "";
  /* a scene whose parked-walk workspace could not be allocated runs on the lane-waiting kernels: report what a launch
   * takes, so that an out-of-memory fallback cannot pass as a measurement of the parked-walk kernels.  What is assumed of
   * the launch itself: sums that fit, a pending-ray pool of full width -- rt_hip_last_launch_kernel() has the fact. */
  bool no_ws;
  {
    std::lock_guard<std::mutex> lock(scene->table_mutex);
    /* (a scene that has met its workspace keeps it, whatever is injected later; one that has not yet would not get it now) */
    no_ws = scene->park_tried ? scene->park_ws == nullptr : (g_fail_alloc.load() & RT_HIP_FAIL_ALLOC_PARK_WS) != 0;
  }
  const PtPickFacts facts = {integrator, 1, 0, !no_ws, !(g_fail_alloc.load() & RT_HIP_FAIL_ALLOC_WIDE_PEND)};
  return pt_kernel_name_of(pt_pick_kernel(scene->view, facts));
}

const char *rt_hip_last_launch_kernel(void) { return pt_kernel_name_of(g_last_kernel); }

int rt_hip_kernel_count(void) { return pt_kernel_count(); }

const char *rt_hip_kernel_launches(int index, uint64_t *launches)
{
  if (index < 0 || index >= pt_kernel_count())
    return nullptr;
  if (launches)
    *launches = pt_kernel_launches(index);
  return pt_kernel_name_of(index);
}

const char *rt_hip_kernel_for_class(const RtHipSceneClass *c)
{
  if (!c)
    return "";
  PtSceneView v;
  memset(&v, 0, sizeof v);
  v.n_spheres = c->n_spheres;
  v.n_meshes = c->n_meshes;
  v.n_triangles = c->n_triangles;
  v.n_bvh_nodes = c->n_triangles ? std::max(1u, c->n_triangles / 8u) : 0u;
  v.any_checker = c->any_checker ? 1u : 0u;
  v.any_refract = c->any_refract ? 1u : 0u;
  v.any_mirror_glass = c->any_mirror_glass ? 1u : 0u;
  v.wide_range = c->wide_range ? 1u : 0u;
  v.mesh_round = c->mesh_round ? 1u : 0u;
  const PtPickFacts facts = {c->integrator, c->samples_per_chunk, c->max_depth, c->have_park_ws != 0, c->wide_pend_ok != 0};
  return pt_kernel_name_of(pt_pick_kernel(v, facts));
}

void rt_hip_selftest_fail_alloc(uint32_t mask) { g_fail_alloc.store(mask); }

int rt_hip_selftest_pool_slots(int device, uint32_t *park_slots_per_xcd, uint32_t *pend_slots_per_xcd)
{
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  DeviceScope scope(device);
  HIP_TRY(scope.status);
  if (park_slots_per_xcd)
    *park_slots_per_xcd = pt_pool_slots_per_xcd(true);
  if (pend_slots_per_xcd)
    *pend_slots_per_xcd = pt_pool_slots_per_xcd(false);
  return RT_HIP_OK;
}

int rt_hip_pool_bytes(int device, size_t *park_ws_bytes, size_t *pend_pool_bytes)
{
  if (device < 0 || device >= 64 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  if (park_ws_bytes)
  {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    const ParkPool &p = g_park[device];
    *park_ws_bytes = p.ws ? park_flag_bytes(p.slots_per_xcd) + (size_t)PT_PARK_XCDS * p.slots_per_xcd * (PT_BLOCK / 64) * (size_t)PT_PARK_WAVE_BYTES : 0;
  }
  if (pend_pool_bytes)
  {
    std::lock_guard<std::mutex> lock(g_pend_mutex);
    const PendPool &p = g_pend[device];
    *pend_pool_bytes = p.ws ? pend_flag_bytes(p.slots_per_xcd) + (size_t)PT_PARK_XCDS * p.slots_per_xcd * p.entries * PT_PEND_FIELDS_HOST * p.columns * sizeof(double) : 0;
  }
  return RT_HIP_OK;
}

int rt_hip_launch_status(int device, uint32_t *flags)
{
  uint32_t f = 0;
  if (flags)
    *flags = 0;
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  DeviceScope scope(device);
  HIP_TRY(scope.status);
  int rc = status_take(device, &f);
  if (rc)
    return rc;
  if (flags)
    *flags = f;
  return status_to_error(f);
}

size_t rt_hip_chunk_workspace_bytes(uint32_t tile_count)
{ /* enough for any scene: the windowed sums of the M_REFRACTION forms are the larger record */
  return (size_t)tile_count * PT_ACC_WS_WORDS_WIN * sizeof(unsigned long long);
}

size_t rt_hip_scene_chunk_workspace_bytes(const RtHipScene *scene, uint32_t tile_count)
{
  const bool windowed = !scene || scene->view.any_refract;
  return (size_t)tile_count * (windowed ? PT_ACC_WS_WORDS_WIN : PT_ACC_WS_WORDS) * sizeof(unsigned long long);
}

uint32_t rt_hip_suggest_chunks(const RtHipScene *scene, uint32_t tile_count, int32_t samples)
{
  return rt_hip_suggest_chunks_depth(scene, tile_count, samples, 0);
}

uint32_t rt_hip_suggest_chunks_depth(const RtHipScene *scene, uint32_t tile_count, int32_t samples, int32_t max_depth)
{
  if (!scene || tile_count == 0 || samples < 1)
    return 1;
  /* scenes with M_REFRACTION: at least as many chunks as the windowed sums need (pt_refr_pool_fits per chunk) */
  uint64_t need = 1;
  if (scene->view.any_refract)
  {
    need = pt_refr_pool_chunks_needed(samples, max_depth);
    if (need == 0 || need > (uint64_t)samples || need * tile_count > 0x7FFFFFFFull)
      need = 1; /* no chunking fits (max_depth > 29): the static kernels, which do not split samples */
  }
  if (samples < 128)
    return (uint32_t)need;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, scene->device) != hipSuccess)
    return (uint32_t)need;
  /* which body the scene takes: the parked-walk kernels render a tile per WAVE (four per workgroup, four workgroups per CU),
   * and a chunk of theirs must be longer -- a wave amortises its walk batches and its final, partly filled walk over its pool */
  bool queued, windowed;
  {
    std::lock_guard<std::mutex> lock(scene->table_mutex);
    const bool have_ws = scene->park_tried ? scene->park_ws != nullptr : (g_fail_alloc.load() & RT_HIP_FAIL_ALLOC_PARK_WS) == 0;
    const PtPickFacts facts = {RT_HIP_TRACE_PATH, (int32_t)(((uint64_t)samples + need - 1) / need), max_depth, have_ws, true};
    const int which = pt_pick_kernel(scene->view, facts);
    queued = pt_kernel_is_queued(which);
    windowed = pt_kernel_is_windowed(which);
  }
  uint64_t want, min_chunk_samples;
  if (queued)
  {
    /* >= 30 rounds of workgroups (the expensive tiles -- those on the mesh -- are few and long: one workgroup of four of them at
     * 4096 spp outlasts a rank's whole ideal share at N = 8), >= 128 samples per chunk.  One rank's share of config 5 at N = 8,
     * ms by chunks (tools/shard_chunks.py, profiles/r05_shard_chunks.txt): 4096 spp 1: 608, 2: 505, 4: 465, 8: 447, 12: 443, 16: 443
     * (ideal 418); 256 spp 1: 39.2, 2: 37.1, 4: 41.8, 8: 42.6 (ideal 27.0).  Round 4's rule (tiles, not workgroups; 64 samples)
     * gave 2 and 4. */
    want = 30ull * 4ull * 4ull * (uint64_t)prop.multiProcessorCount;
  }
  else
  {
    /* aim for >= 20 workgroups per resident slot (5 per CU), so the last, partly filled round
     * of the launch is a small fraction of it.  (One rank's share of the headline frame at
     * N = 8 / 4 / 2, ms by chunks: 2: 30.0, 4: 29.3, 6: 29.4, 8: 29.6, 16: 30.9 / 1: 59.1, 2: 57.7, 4: 57.5 / 1: 114.7, 2: 113.4.) */
    want = 20ull * 5ull * (uint64_t)prop.multiProcessorCount;
  }
  /* a chunk keeps >= 128 samples (a workgroup's fixed costs -- staging, keys, culling, the resolve pass -- against its pool: config 3's
   * share at N = 8, 256 spp, ms by chunks 1: 2.63, 2: 2.61, 4: 2.70, 8: 3.01); the M_REFRACTION forms >= 64 (a refractive sample
   * is two to three times the rays: the glass mesh's share at N = 8, 256 spp 2: 48.8, 4: 45.9, 8: 46.9) */
  min_chunk_samples = windowed ? 64 : 128;
  uint64_t chunks = (want + tile_count - 1) / tile_count;
  const uint64_t cap = (uint64_t)samples / min_chunk_samples;
  if (chunks > cap) chunks = cap;
  if (chunks > 16) chunks = 16;
  if (chunks < need) chunks = need;
  return chunks < 1 ? 1u : (uint32_t)chunks;
}

/* bvh_probe's bounding sphere of the triangles for one near_R: the thresholds of a bounding entry of the flat
 * filter in its compare form, widened exactly as pt_build_filter widens them (e = 2^-24, A = |c| + near_R:
 * |tca32 - tca| <= 6.2 e A, |d2_32 - d2| <= 20.5 e A^2 for origins within near_R; the conversions to fp32 are
 * inside the (1 + k e) factors).  Non-finite or overflowing values give thresholds that keep every ray. */
static void mesh_bound_for(const RtHipScene *scene, double near_R, float out[5])
{
  const float inf = std::numeric_limits<float>::infinity();
  out[0] = out[1] = out[2] = 0.f;
  out[3] = inf;
  out[4] = -inf;
  if (!(scene->mesh_R >= 0) || !(scene->mesh_R < 1e18) || !(scene->mesh_c_norm < 1e18) || !(near_R < 1e18))
    return;
  const double e = 5.9604644775390625e-08, A = scene->mesh_c_norm + near_R;
  for (int a = 0; a < 3; a++)
    out[a] = (float)scene->mesh_c[a];
  out[3] = (float)((scene->mesh_R * scene->mesh_R + 32.0 * e * A * A) * (1.0 + 8.0 * e));
  out[4] = -(float)((scene->mesh_R + 10.0 * e * A) * (1.0 + 4.0 * e));
}

/* BigPrune (pt_filter.h, where the bounds are derived): for one near_R, the distance margin delta, the least estimate tmin
 * and per sphere the least q32 of a leading wall-sized sphere that may prune the others.  With e = 2^-24, A = |c| + near_R +
 * tol, W >= r2_hi' - r^2 (pt_sign_widen_total of pt_device.h, the very function pt_build_filter widens by, x 1.001), E = 28 e A^2 + 6 e | |c|^2 -
 * r^2 |:  qmin = (r / 16)^2 + W + E,  tmin = 2 (tol + 11.2 e A),  delta = 1.5 max (22.4 e A + 8 (W + E) / r).
 * Anything non-finite or implausible switches the pruning off. */
static void big_prune_for(const RtHipScene *scene, double near_R, double filt_shift, PtLaunch &L)
{
  L.big_pairs = 0;
  L.big_delta = L.big_tmin = 0.f;
  for (int k = 0; k < 8; k++)
    L.big_qmin[k] = std::numeric_limits<float>::infinity();
#ifdef PT_DEV_KERNELS
  static const bool off = [] {
    const char *e = getenv("RT_HIP_NO_BIG_PRUNE"); /* development switch (A/B) */
    return e && e[0] == '1';
  }();
#else
  const bool off = false;
#endif
  /* the kernels whose sphere filter is the sign-test form from LDS: sphere-only small scenes, and hierarchy scenes whose
   * spheres fit the staging (the parked-walk kernels filter the spheres alone) */
  const bool sign_form = !scene->view.wide_range &&
                         (pt_geom_in_lds(scene->view) ? (scene->view.n_triangles == 0 ? pt_filter_in_lds(scene->view) : !pt_filter_in_lds(scene->view))
                                                      : (scene->view.n_triangles == 0)); /* pt_render_tiles_pool_mem_s, pt_render_tiles_refr_pool_mem
                                                                             * (scenes it takes by preference, pt_prefer_streaming, satisfy the first arm) */
  if (off || scene->n_big < 2 || !sign_form)
    return;
  const double e = 5.9604644775390625e-08, f = 1.0 / 16.0;
  double delta = 0, tmin = 0;
  for (int k = 0; k < scene->n_big; k++)
  {
    const double r = scene->big_r[k], c = scene->big_c[k], A = c + near_R + filt_shift;
    const double g = std::fabs(c * c - r * r);
    /* W >= r2_hi' - r^2 of the table: pt_build_filter widens by pt_sign_widen_total(A_table, g) with A_table = |c| + near_R <= A
     * (the function is increasing in A), then rounds kq DOWN to fp32 (at most 2 e |kq| more: inside the factor 1.001) */
    const double W = pt_sign_widen_total(A, g) * 1.001, E = 28.0 * e * A * A + 6.0 * e * g;
    const double qmin = f * f * r * r + W + E;
    const double bias = (W + E) / (2.0 * f * r);
    if (!(qmin < 1e30) || !(bias < 1e3))
      return;
    L.big_qmin[k] = (float)(qmin * (1.0 + 4.0 * e));
    delta = std::fmax(delta, 1.5 * (22.4 * e * A + bias));
    tmin = std::fmax(tmin, 2.0 * (filt_shift * 1.0001 + 11.2 * e * A));
  }
  L.big_delta = (float)(delta * (1.0 + 4.0 * e));
  L.big_tmin = (float)(tmin * (1.0 + 4.0 * e));
  L.big_pairs = (uint32_t)scene->n_big / 2u;
}

/* How far on the outer side of a hull facet F (pt_build_hull_flags) a ray must point, mu < m . d, to be certain
 * not to meet a triangle.  With u = 2^-53, sigma = |e1||e2| / |e1 x e2| <= 16 (the flag's shape limit), D >= |o - v0|
 * and the hit distance (3 (near_R + extent) covers both):
 *   - the computed hit point on F lies within delta of F's plane: N . (o + t d - v0) = e2 . q - t a exactly (N = e1 x e2,
 *     q = s x e1, a = e1 . (d x e2): intersect_triangle's own quantities), which vanishes for the exact t; the computed
 *     t carries 6 u (|s| + t) |e1||e2| / |N| + 3 u |s| of plane distance, the point's own three roundings 6 u D more:
 *     delta <= 26 u sigma D <= 1.4e-13 (near_R + extent);
 *   - every triangle point p has m . (p - v0) <= tau = 2^-43 extent;
 *   - so a hit needs t <= (tau + delta) / mu, and mu = 4 (tau + delta) / EPSILON puts that at EPSILON / 4, where the
 *     exact test (t > EPSILON, raytracer.c:150) rejects it, its own rounding of t (relative ~1e-12) included.
 * Never below 1e-3; scenes so large that mu reaches 1 simply never skip a walk.  D holds for rays whose hit
 * distance is at most 2 near_R: then |o - v0| <= 1.0001 x that + the facet's size as well; trace_step drops the
 * facet's mark for any other (a bounce that came in from a far point of a wall-sized sphere, say). */
static double hull_margin_for(const RtHipScene *scene, double near_R)
{
  if (!scene->hull_flags || !(near_R < 1e150))
    return 2.0; /* no ray has m . d > 2 */
  const double extent = scene->mesh_c_norm + scene->mesh_R;
  const double tau = 1.1368683772161603e-13 * extent, delta = 1.4e-13 * (near_R + extent);
  return std::fmax(1e-3, 4.0 * (tau + delta) / 1e-8);
}

int rt_hip_render_tiles(const RtHipScene *scene, const RtHipCamera *camera, const RtHipParams *params,
                        float *d_tiles_rgb, uint8_t *d_tiles_rgb8, uint64_t *d_stats, void *stream)
{
  return rt_hip_render_tiles_chunked(scene, camera, params, 1, nullptr, d_tiles_rgb, d_tiles_rgb8, d_stats, stream);
}

int rt_hip_render_tiles_chunked(const RtHipScene *scene, const RtHipCamera *camera, const RtHipParams *params,
                                uint32_t sample_chunks, void *d_workspace, float *d_tiles_rgb,
                                uint8_t *d_tiles_rgb8, uint64_t *d_stats, void *stream)
{
  if (!scene || !camera || !d_tiles_rgb)
    return fail(RT_HIP_EINVAL, "scene, camera and d_tiles_rgb are required");
  if (sample_chunks < 1 || (params && (int64_t)sample_chunks > params->samples))
    return fail(RT_HIP_EINVAL, "sample_chunks must be in [1, samples]");
  if (sample_chunks > 1 && !d_workspace)
    return fail(RT_HIP_EINVAL, "sample_chunks > 1 needs a workspace of rt_hip_chunk_workspace_bytes(tile_count)");
  int rc = check_params(params);
  if (rc)
    return rc;
  const bool cast_ray = params->integrator == RT_HIP_CAST_RAY;
  if (!cast_ray && scene->view.any_refract && params->max_depth > PT_REFRACT_MAX_DEPTH)
    return fail(RT_HIP_ELIMIT, "scenes with M_REFRACTION materials support max_depth <= %d (two rays per "
                               "refractive hit, raytracer.c:523-529; the pending-ray stack is fixed)",
                PT_REFRACT_MAX_DEPTH);
  if (cast_ray && scene->any_mirror_glass && params->max_depth > PT_REFRACT_MAX_DEPTH)
    return fail(RT_HIP_ELIMIT, "cast_ray with M_REFLECTION|M_REFRACTION materials supports max_depth <= %d (two "
                               "rays per such hit, raytracer.c:609-628; the pending-ray stack is fixed)",
                PT_REFRACT_MAX_DEPTH);
  const uint32_t tx = tiles_x_of(params->width), ty = tiles_y_of(params->height);
  const uint64_t n_tiles = (uint64_t)tx * ty;
  if (params->tile_count == 0)
    return RT_HIP_OK;
  if (params->tile_stride == 0 && params->tile_count > 1)
    return fail(RT_HIP_EINVAL, "tile_stride must be >= 1");
  const uint64_t last = (uint64_t)params->tile_first + (uint64_t)(params->tile_count - 1) * params->tile_stride;
  if (last >= n_tiles)
    return fail(RT_HIP_EINVAL, "tile range [%u + k*%u, k < %u] exceeds the image's %llu tiles", params->tile_first,
                params->tile_stride, params->tile_count, (unsigned long long)n_tiles);

  PtLaunch L;
  memset(&L, 0, sizeof L);
  L.scene = scene->view;
  memcpy(L.cam.pos, camera->position, sizeof L.cam.pos);
  memcpy(L.cam.horizontal, camera->horizontal, sizeof L.cam.horizontal);
  memcpy(L.cam.vertical, camera->vertical, sizeof L.cam.vertical);
  memcpy(L.cam.llc, camera->lower_left_corner, sizeof L.cam.llc);
  L.width = params->width;
  L.height = params->height;
  L.samples = params->samples;
  L.max_depth = params->max_depth;
  L.seed = params->seed;
  {
    /* Ray origins are the camera or points on primitives.  The packed-fp32 filter of
     * scan_filtered is built for origins within near_R; a ray starting farther out (e.g. on
     * the far side of a radius-1e4 "wall" sphere) is still traced exactly, it just skips
     * the filter.  near_R only trades filter tightness against that fallback. */
    const double *c = camera->position;
    const double cam = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    L.near_R = 1.5 * (cam + scene->reach) + 1.0;
    if (!(L.near_R < RT_NEAR_R_LIMIT))
      return fail(RT_HIP_EINVAL, "scene extent %g is not a usable finite bound", L.near_R);
    L.near_R2 = L.near_R * L.near_R;
    L.filt_shift = 12.0 * 5.9604644775390625e-08 * (scene->max_center + L.near_R) * (1.0 + 1e-9);
    mesh_bound_for(scene, L.near_R, L.mesh_bound);
    big_prune_for(scene, L.near_R, L.filt_shift, L);
    L.hull_margin = hull_margin_for(scene, L.near_R);
#ifdef PT_DIAG
    { /* the diagnostic build only: walk the rays the probe or the hull rule would not walk, and count any that find a triangle */
      const char *flag = getenv("RT_HIP_DIAG_WALK_REJECTED");
      L.diag_flags = (flag && flag[0] == '1') ? 1u : 0u;
    }
#endif
    L.background = 10 / 255.0;
    L.t_start = 1.7976931348623157e308; /* DBL_MAX */
    L.w_minus_1 = (double)params->width - 1.0;
    L.h_minus_1 = (double)params->height - 1.0;
    L.inv_w_minus_1 = 1.0 / L.w_minus_1; /* IEEE division on the host: correctly rounded */
    L.inv_h_minus_1 = 1.0 / L.h_minus_1;
  }
  {
    /* Fixed-point scale of the per-pixel sums (pt_render_tiles): a sample's radiance is
     * sum_k T_k (.) e_k with throughput T <= 1 (albedo/prob <= 1, cos <= 1) over at most
     * max_depth + 2 events, so every term and every sample is bounded by
     * (max_depth + 2) * max(BACKGROUND, max emission).  Two conditions on the power-of-two scale: the sum
     * of `samples` samples stays below 2^62, and a single term stays below 2^51 -- the kernels read a
     * term's integer off an fp64 mantissa (fixed_term in pt_math.h). */
    const double per_sample = ((double)params->max_depth + 2.0) * std::fmax(10.0 / 255.0, scene->max_emission) * 1.01;
    const double bound = per_sample * (double)params->samples;
    if (!(bound > 0) || !(bound < 1e300))
      return fail(RT_HIP_EINVAL, "emission magnitudes give no finite radiance bound (%g)", bound);
    int e = 0, e1 = 0;
    (void)std::frexp(4611686018427387904.0 / bound, &e);      /* 2^62 / bound = m * 2^e, m in [0.5, 1) */
    (void)std::frexp(2251799813685248.0 / per_sample, &e1);   /* 2^51 / per_sample */
    if (e1 < e)
      e = e1;
    L.acc_scale = std::ldexp(1.0, e - 1);
    L.acc_inv_scale = std::ldexp(1.0, 1 - e);
  }
  L.tile_first = params->tile_first;
  L.tile_stride = params->tile_stride;
  L.tile_count = params->tile_count;
  L.tiles_x = tx;
  L.integrator = cast_ray ? 1u : 0u;
  L.acc_ws = static_cast<unsigned long long *>(d_workspace);
  L.tiles_rgb = d_tiles_rgb;
  L.tiles_rgb8 = d_tiles_rgb8;
  L.stats = reinterpret_cast<unsigned long long *>(d_stats);

  DeviceScope scope(scene->device);
  HIP_TRY(scope.status);
  rc = status_word_for(scene->device, &L.status);
  if (rc)
    return rc;
  if (scene->view.n_bvh_nodes != 0 && !cast_ray)
  {
    /* parked-walk workspace: the device's shared pool (flags + rings).  Without it the pick table's park = NO rows apply
     * (the lane-waiting kernels), and rt_hip_kernel_name reports those. */
    std::lock_guard<std::mutex> lock(scene->table_mutex);
    if (!scene->park_tried)
    {
      scene->park_tried = true;
      scene->park_ws = park_acquire_ws(scene->device, &scene->park_slots_per_xcd);
    }
    if (scene->park_ws)
    {
      L.park_flags = reinterpret_cast<uint32_t *>(scene->park_ws);
      L.park_ws = scene->park_ws + park_flag_bytes(scene->park_slots_per_xcd);
      L.park_slots_per_xcd = scene->park_slots_per_xcd;
    }
  }
  /* ---- which kernel (pt_kernel.hip: pt_pick_table), and how many sample chunks it takes ---- */
  /* scenes with M_REFRACTION: the windowed sums of the pooled / parked-walk forms hold a bounded number of samples per chunk
   * (pt_refr_pool_fits).  A caller that handed over a workspace gets at least as many chunks as that needs -- the image does
   * not depend on the chunk count, and the workspace's size does not either; without a workspace the launch keeps its one
   * chunk, and where that does not fit the table's fit = NO row (the static kernel of the family) renders it. */
  if (!cast_ray && scene->view.any_refract && d_workspace)
  {
    const uint64_t need = pt_refr_pool_chunks_needed(params->samples, params->max_depth);
    if (need > sample_chunks && need <= (uint64_t)params->samples && need * params->tile_count <= 0x7FFFFFFFull)
      sample_chunks = (uint32_t)need;
  }
  const int32_t samples_per_chunk = (int32_t)(((int64_t)params->samples + sample_chunks - 1) / sample_chunks);
  PtPickFacts facts = {L.integrator, samples_per_chunk, params->max_depth, L.park_ws != nullptr, true};
  int which = pt_pick_kernel(L.scene, facts);
  auto chunks_of = [&](int k) {
    if (pt_kernel_takes_chunks(k))
      return sample_chunks;
    /* the static bodies (cast_ray, the fit = NO rows of M_REFRACTION) do not split samples */
    return 1u;
  };
  L.sample_chunks = chunks_of(which);
  L.acc_windows = pt_kernel_is_windowed(which) ? 1u : 0u;
  if ((uint64_t)L.tile_count * L.sample_chunks > 0x7FFFFFFFull)
    return fail(RT_HIP_EINVAL, "tile_count x sample_chunks exceeds the grid limit");
  size_t slot = 0;
  rc = acquire_tables(scene, L.near_R, static_cast<hipStream_t>(stream), &L.scene.filt, &L.scene.bvh_nodes, &slot);
  if (rc)
    return rc;
  hipError_t e;
  if (pt_kernel_uses_pend_pool(which))
  {
    std::lock_guard<std::mutex> pend_lock(g_pend_mutex);
    rc = pend_pool_for(scene->device, (uint32_t)params->max_depth + 2u, pt_kernel_pend_columns_of(which), L);
    /* the parked-walk refraction kernels want four times the stacks per slot (1.2 GB at depth 5, 5.7 GB at 32): where that
     * cannot be had, the pool of the other kernels will do -- the table's fit = NO row names the static kernel of the family */
    if (rc == RT_HIP_ENOMEM && pt_kernel_pend_columns_of(which) > PT_PEND_COLUMNS)
    {
      facts.wide_pend_ok = false;
      which = pt_pick_kernel(L.scene, facts);
      L.sample_chunks = chunks_of(which);
      L.acc_windows = pt_kernel_is_windowed(which) ? 1u : 0u;
      rc = pt_kernel_uses_pend_pool(which) ? pend_pool_for(scene->device, (uint32_t)params->max_depth + 2u, pt_kernel_pend_columns_of(which), L)
                                           : RT_HIP_OK;
    }
    if (rc)
    {
      release_tables(scene, slot, static_cast<hipStream_t>(stream));
      return rc;
    }
    e = pt_launch_render(L, static_cast<hipStream_t>(stream), which);
  }
  else
    e = pt_launch_render(L, static_cast<hipStream_t>(stream), which);
  release_tables(scene, slot, static_cast<hipStream_t>(stream));
  if (e != hipSuccess)
    return fail(RT_HIP_ERUNTIME, "%s launch: %s", pt_kernel_name_of(which), hipGetErrorString(e));
  g_last_kernel = which;
  return RT_HIP_OK;
}

int rt_hip_selftest_math(int op, const double *h_a, const double *h_b, double *h_out, size_t n, int device)
{
  if (!h_a || !h_b || !h_out || op < 0 || op > 8 || (op == 8 && n < 8))
    return fail(RT_HIP_EINVAL, "bad self-test arguments");
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  if (n == 0)
    return RT_HIP_OK;
  DeviceScope scope(device);
  HIP_TRY(scope.status);
  double *d = nullptr;
  hipError_t e = hipMalloc(&d, 3 * n * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d, h_a, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + n, h_b, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d + 2 * n, 0, n * sizeof(double)); /* (op 8 accumulates into out) */
  if (e == hipSuccess) e = pt_launch_selftest(op, d, d + n, d + 2 * n, n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(h_out, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess)
    return fail(RT_HIP_ERUNTIME, "self-test: %s", hipGetErrorString(e));
  return RT_HIP_OK;
}

int rt_hip_selftest_xcc(uint32_t n_workgroups, uint32_t h_counts[16], int device)
{
  if (!h_counts || n_workgroups == 0 || n_workgroups > (1u << 20))
    return fail(RT_HIP_EINVAL, "bad self-test arguments");
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  DeviceScope scope(device);
  HIP_TRY(scope.status);
  unsigned int *d = nullptr;
  hipError_t e = hipMalloc(&d, 16 * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemset(d, 0, 16 * sizeof(unsigned int));
  if (e == hipSuccess) e = pt_launch_selftest_xcc(d, n_workgroups, nullptr);
  if (e == hipSuccess) e = hipMemcpy(h_counts, d, 16 * sizeof(unsigned int), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess)
    return fail(RT_HIP_ERUNTIME, "xcc self-test: %s", hipGetErrorString(e));
  return RT_HIP_OK;
}

int rt_hip_selftest_intersect(int kind, const double *h_rays, const double *h_prims, size_t n, double near_R,
                              uint8_t *h_hit, double *h_tuv, uint64_t *h_keep, int device)
{
  if ((kind != 0 && kind != 1) || !h_rays || !h_prims || !h_hit || !h_tuv || !h_keep)
    return fail(RT_HIP_EINVAL, "bad self-test arguments");
  if (!(near_R > 0) || !(near_R < 1e15) || n > 0x7FFFFFFFu)
    return fail(RT_HIP_EINVAL, "near_R must be a positive finite bound, n < 2^31");
  if (device < 0 || device >= usable_devices())
    return fail(RT_HIP_ENODEV, "no HIP device %d", device);
  if (n == 0)
    return RT_HIP_OK;
  try
  {
    /* the records exactly as rt_hip_scene_create lays them out (same helpers) */
    const size_t rec = kind == 0 ? 4 : 9;
    std::vector<double> prims(rec * n), entry(PT_ENTRY_SRC_STRIDE * n);
    double max_center = 0;
    for (size_t i = 0; i < n; i++)
    {
      double *e = &entry[PT_ENTRY_SRC_STRIDE * i];
      if (kind == 0)
      {
        const double *p = h_prims + 4 * i;
        if (!(std::fabs(p[3]) >= 1e-100) || !(std::fabs(p[3]) <= 1e17))
          return fail(RT_HIP_ELIMIT, "sphere %zu: |radius| %g outside [1e-100, 1e17]", i, p[3]);
        sphere_entry(p, p[3], e);
        memcpy(&prims[4 * i], e, 4 * sizeof(double));
      }
      else
        triangle_entry(h_prims + 9 * i, h_prims + 9 * i + 3, h_prims + 9 * i + 6, &prims[9 * i], e);
      if (!(e[4] <= 1e17))
        return fail(RT_HIP_ELIMIT, "primitive %zu: centre beyond 1e17", i);
      max_center = std::fmax(max_center, e[4]);
    }
    const double filt_shift = 12.0 * 5.9604644775390625e-08 * (max_center + near_R) * (1.0 + 1e-9); /* as rt_hip_render_tiles */
    const size_t n_blocks = (n + 63) / 64;
    const size_t filt_bytes = (n_blocks * 32 + 1) * (size_t)PT_FILT_STRIDE * 2 * sizeof(float);
    const size_t b_rays = 6 * n * 8, b_prims = rec * n * 8, b_entry = entry.size() * 8, b_tuv = 3 * n * 8, b_keep = 3 * n * 8;
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_tri32 = kind == 1 ? n * PT_TRI32_STRIDE * sizeof(float) : 0;
    const size_t o_rays = 0, o_prims = o_rays + pad(b_rays), o_entry = o_prims + pad(b_prims), o_filt = o_entry + pad(b_entry),
                 o_tri32 = o_filt + pad(filt_bytes), o_tuv = o_tri32 + pad(b_tri32), o_keep = o_tuv + pad(b_tuv),
                 o_hit = o_keep + pad(b_keep), total = o_hit + pad(n);
    DeviceScope scope(device);
    HIP_TRY(scope.status);
    char *d = nullptr;
    hipError_t e = hipMalloc(&d, total);
    if (e != hipSuccess)
      return fail(RT_HIP_ENOMEM, "hipMalloc(%zu): %s", total, hipGetErrorString(e));
    e = hipMemset(d + o_filt, 0, filt_bytes);
    if (e == hipSuccess) e = hipMemcpy(d + o_rays, h_rays, b_rays, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_prims, prims.data(), b_prims, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_entry, entry.data(), b_entry, hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = pt_launch_selftest_intersect(kind, reinterpret_cast<double *>(d + o_rays), reinterpret_cast<double *>(d + o_prims),
                                       reinterpret_cast<double *>(d + o_entry), reinterpret_cast<float *>(d + o_filt),
                                       reinterpret_cast<float *>(d + o_tri32), (uint32_t)n, near_R, filt_shift, reinterpret_cast<uint8_t *>(d + o_hit),
                                       reinterpret_cast<double *>(d + o_tuv), reinterpret_cast<unsigned long long *>(d + o_keep),
                                       nullptr);
    if (e == hipSuccess) e = hipMemcpy(h_hit, d + o_hit, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_tuv, d + o_tuv, b_tuv, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_keep, d + o_keep, b_keep, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess)
      return fail(RT_HIP_ERUNTIME, "intersect self-test: %s", hipGetErrorString(e));
    return RT_HIP_OK;
  }
  catch (const std::bad_alloc &)
  {
    return fail(RT_HIP_ENOMEM, "host allocation failed in rt_hip_selftest_intersect");
  }
}

int rt_hip_untile(const float *d_tiles_rgb, const uint8_t *d_tiles_rgb8, int32_t width, int32_t height,
                  uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count, float *d_image_rgb,
                  uint8_t *d_image_rgb8, void *stream)
{
  if (width < 1 || height < 1)
    return fail(RT_HIP_EINVAL, "bad image size");
  if ((d_image_rgb && !d_tiles_rgb) || (d_image_rgb8 && !d_tiles_rgb8))
    return fail(RT_HIP_EINVAL, "an output image needs its tile buffer");
  if (tile_count == 0 || (!d_image_rgb && !d_image_rgb8))
    return RT_HIP_OK;
  const uint64_t n_tiles = (uint64_t)tiles_x_of(width) * tiles_y_of(height);
  if ((uint64_t)tile_first + (uint64_t)(tile_count - 1) * tile_stride >= n_tiles)
    return fail(RT_HIP_EINVAL, "tile range exceeds the image");
  hipError_t e = pt_launch_untile(d_tiles_rgb, d_tiles_rgb8, width, height, tile_first, tile_stride, tile_count,
                                  d_image_rgb, d_image_rgb8, static_cast<hipStream_t>(stream));
  if (e != hipSuccess)
    return fail(RT_HIP_ERUNTIME, "pt_untile launch: %s", hipGetErrorString(e));
  return RT_HIP_OK;
}

/* Whole image on n_devices GPUs of this process.  Device g renders tiles
 * g, g+G, g+2G, ... into its own compact buffer; the buffers are gathered on
 * device 0 with grouped ncclSend/ncclRecv (point-to-point over xGMI: a gather
 * to one root uses the root's 7 direct links concurrently, there is no ring),
 * scattered to the row-major image there, and copied to the host. */
} /* extern "C" */

namespace
{

/* Everything rt_hip_render_image() needs between calls -- per device: the uploaded scene, a
 * stream, timing events, the compact tile buffers, counters, the chunk workspace; on device 0 the
 * gathered tiles and the row-major images; and the RCCL communicators -- is kept in one cached
 * context and reused while the device count, the image size and the scene's bytes stay the same
 * (an animation loop calling render() per frame re-creates nothing; ncclCommInitAll alone costs
 * tens of milliseconds per call at 8 devices).  rt_hip_release_cache() drops it. */
struct ImageCtx
{
  struct Dev
  {
    RtHipScene *scene = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    float *tiles = nullptr;
    uint8_t *tiles8 = nullptr;
    uint64_t *stats = nullptr;
    void *ws = nullptr;
    uint32_t count = 0;
  };
  int G = 0, W = 0, H = 0;
  /* logical device g runs on physical device phys[g] (rt_hip_set_device_map; the identity without a map).  Logical devices
   * that share a physical one each keep their own scene, stream and tile buffers -- everything above the gather is the same
   * code as with G distinct GPUs.  comm_of[g]: index of phys[g] among the DISTINCT physical devices = its RCCL rank (RCCL
   * refuses two ranks on one device); lead[g]: the first logical device on phys[g], whose stream carries that rank's sends. */
  std::vector<int> phys, comm_of, lead;
  int n_phys = 0;
  bool force_comm = false; /* RT_HIP_FORCE_COMM=1: communicators and the gather's send/recv block even with one device */
  std::vector<unsigned char> scene_bytes; /* the scene this context was built for (scene_walk's runs): compared run by run */
  std::vector<Dev> dev;
  std::vector<ncclComm_t> comms; /* one per distinct physical device (comm_of) */
  std::vector<hipEvent_t> done;  /* per logical device: its tiles are complete (what a same-device copy or the lead's sends wait for) */
  float *all_tiles = nullptr, *image = nullptr;
  uint8_t *all_tiles8 = nullptr, *image8 = nullptr;
  uint64_t builds = 0; /* how many times a context was (re)built: exposed for tests */
};
ImageCtx g_ctx;
std::mutex g_ctx_mutex;
double g_last_phases[3] = {0, 0, 0}; /* rt_hip_last_image_phases */

/* The logical -> physical device map of rt_hip_render_image (rt_hip_set_device_map, or RT_HIP_DEVICE_MAP=0,0,1 read at
 * the first frame).  Empty = the identity.  Guarded by g_ctx_mutex. */
std::vector<int> g_device_map;
bool g_device_map_env_read = false;

void device_map_from_env()
{
  if (g_device_map_env_read)
    return;
  g_device_map_env_read = true;
  const char *e = getenv("RT_HIP_DEVICE_MAP");
  if (!e || !*e || !g_device_map.empty())
    return;
  std::vector<int> m;
  for (const char *p = e; *p;)
  {
    char *end = nullptr;
    const long v = strtol(p, &end, 10);
    if (end == p || v < 0 || v > 63)
      return; /* malformed: ignored as a whole */
    m.push_back((int)v);
    p = end;
    if (*p == ',')
      p++;
    else if (*p)
      return;
  }
  g_device_map = m;
}

void ctx_release(ImageCtx &c)
{
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (int g = 0; g < (int)c.dev.size(); g++)
  {
    (void)hipSetDevice(g < (int)c.phys.size() ? c.phys[g] : g);
    ImageCtx::Dev &d = c.dev[g];
    if (d.stream) (void)hipStreamSynchronize(d.stream);
  }
  for (int r = 0; r < (int)c.comms.size(); r++)
    if (c.comms[r])
    {
      for (int g = 0; g < (int)c.comm_of.size(); g++)
        if (c.comm_of[g] == r)
        {
          (void)hipSetDevice(c.phys[g]);
          break;
        }
      (void)ncclCommDestroy(c.comms[r]);
    }
  for (int g = 0; g < (int)c.dev.size(); g++)
  {
    (void)hipSetDevice(g < (int)c.phys.size() ? c.phys[g] : g);
    ImageCtx::Dev &d = c.dev[g];
    if (d.t0) (void)hipEventDestroy(d.t0);
    if (d.t1) (void)hipEventDestroy(d.t1);
    if (g < (int)c.done.size() && c.done[g]) (void)hipEventDestroy(c.done[g]);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    (void)hipFree(d.tiles);
    (void)hipFree(d.tiles8);
    (void)hipFree(d.stats);
    (void)hipFree(d.ws);
    rt_hip_scene_destroy(d.scene);
  }
  if (!c.dev.empty())
  {
    (void)hipSetDevice(c.phys.empty() ? 0 : c.phys[0]);
    (void)hipFree(c.all_tiles);
    (void)hipFree(c.all_tiles8);
    (void)hipFree(c.image);
    (void)hipFree(c.image8);
  }
  (void)hipSetDevice(prev);
  const uint64_t builds = c.builds;
  c = ImageCtx();
  c.builds = builds;
}

/* Everything that defines the scene, field by field (struct padding is not part of the scene), as runs of bytes handed
 * to `f(ptr, n)` in a fixed order.  The cached context keeps the concatenation (ImageCtx::scene_bytes) and is reused only
 * while a call's scene equals it byte for byte: scene_matches() compares run by run (memcmp, stopping at the first
 * difference) against the kept copy -- no per-call serialisation, no hashing: a frame of config 5's mesh used to rebuild
 * and hash 1.2 MB on the host inside every rt_hip_render_image() call (round-3 advisor finding). */
template <class F>
void scene_walk(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes, size_t n_meshes, F &&f)
{
  f(&n_spheres, sizeof n_spheres);
  f(&n_meshes, sizeof n_meshes);
  for (size_t i = 0; i < n_spheres; i++)
  {
    f(&spheres[i].flags, sizeof spheres[i].flags);
    f(&spheres[i].radius, sizeof(double) * 10); /* radius, center, color, emission are contiguous doubles */
  }
  for (size_t m = 0; m < n_meshes; m++)
  {
    f(&meshes[m].flags, sizeof meshes[m].flags);
    f(meshes[m].color, sizeof(double) * 6); /* color, emission */
    f(&meshes[m].num_triangles, sizeof meshes[m].num_triangles);
    if (meshes[m].vertices)
      f(meshes[m].vertices, meshes[m].num_triangles * 3 * sizeof(RtHipVertex));
  }
}

bool scene_matches(const std::vector<unsigned char> &kept, const RtHipSphere *spheres, size_t n_spheres,
                   const RtHipMesh *meshes, size_t n_meshes)
{
  size_t at = 0;
  bool same = true;
  scene_walk(spheres, n_spheres, meshes, n_meshes, [&](const void *p, size_t n) {
    if (!same)
      return;
    if (n > kept.size() - at || memcmp(kept.data() + at, p, n) != 0)
      same = false;
    else
      at += n;
  });
  return same && at == kept.size();
}

void scene_serialise(std::vector<unsigned char> &bytes, const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes,
                     size_t n_meshes)
{
  bytes.clear();
  scene_walk(spheres, n_spheres, meshes, n_meshes, [&](const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    bytes.insert(bytes.end(), b, b + n);
  });
}

#define CTX_TRY(expr)                                                                               \
  do                                                                                                \
  {                                                                                                 \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
    {                                                                                               \
      int c_ = fail(e_ == hipErrorOutOfMemory ? RT_HIP_ENOMEM : RT_HIP_ERUNTIME, "%s: %s", #expr,   \
                    hipGetErrorString(e_));                                                         \
      ctx_release(g_ctx);                                                                           \
      (void)hipSetDevice(prev);                                                                     \
      return c_;                                                                                    \
    }                                                                                               \
  } while (0)

/* makes g_ctx fit this call (device count and map, image size, scene); caller holds g_ctx_mutex */
int ctx_prepare(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes, size_t n_meshes, int G,
                const std::vector<int> &phys, int W, int H, int prev)
{
  /* RT_HIP_FORCE_COMM=1: build the RCCL communicator(s) and run the gather's grouped send / recv block even where no
   * tile has to change devices (one device, or logical devices that all share one): every segment then travels through
   * RCCL as a send to self -- how a one-GPU box exercises library load, bootstrap, communicator creation / destruction
   * and the send / recv kernels of the N > 1 path */
  const char *fc = getenv("RT_HIP_FORCE_COMM");
  const bool force_comm = fc && fc[0] == '1';
  ImageCtx &c = g_ctx;
  if (c.G == G && c.W == W && c.H == H && c.force_comm == force_comm && (int)c.dev.size() == G && c.phys == phys &&
      scene_matches(c.scene_bytes, spheres, n_spheres, meshes, n_meshes))
    return RT_HIP_OK;
  ctx_release(c);
  c.builds++;
  c.dev.resize(G);
  c.phys = phys;
  c.comm_of.assign(G, 0);
  c.lead.assign(G, 0);
  c.done.assign(G, nullptr);
  c.n_phys = 0;
  for (int g = 0; g < G; g++)
  {
    int first = g;
    for (int j = 0; j < g; j++)
      if (phys[j] == phys[g])
      {
        first = j;
        break;
      }
    c.lead[g] = first;
    c.comm_of[g] = first == g ? c.n_phys++ : c.comm_of[first];
  }
  c.comms.assign(c.n_phys, nullptr);
  const uint32_t n_tiles = tiles_x_of(W) * tiles_y_of(H);
  const size_t n_px = (size_t)W * H;
  for (int g = 0; g < G; g++)
  {
    ImageCtx::Dev &d = c.dev[g];
    d.count = (n_tiles > (uint32_t)g) ? (n_tiles - g + G - 1) / G : 0;
    int rc = rt_hip_scene_create(spheres, n_spheres, meshes, n_meshes, phys[g], &d.scene);
    if (rc)
    {
      ctx_release(c);
      (void)hipSetDevice(prev);
      return rc;
    }
    CTX_TRY(hipSetDevice(phys[g]));
    CTX_TRY(hipStreamCreate(&d.stream));
    CTX_TRY(hipEventCreate(&d.t0));
    CTX_TRY(hipEventCreate(&d.t1));
    CTX_TRY(hipEventCreateWithFlags(&c.done[g], hipEventDisableTiming));
    const size_t slots = d.count ? d.count : 1;
    CTX_TRY(hipMalloc(&d.tiles, slots * 192 * sizeof(float)));
    CTX_TRY(hipMalloc(&d.tiles8, slots * 192));
    CTX_TRY(hipMalloc(&d.stats, RT_HIP_NSTATS * sizeof(uint64_t)));
  }
  CTX_TRY(hipSetDevice(phys[0]));
  CTX_TRY(hipMalloc(&c.image, n_px * 3 * sizeof(float)));
  CTX_TRY(hipMalloc(&c.image8, n_px * 3));
  if (G > 1 || force_comm)
  {
    CTX_TRY(hipMalloc(&c.all_tiles, (size_t)n_tiles * 192 * sizeof(float)));
    CTX_TRY(hipMalloc(&c.all_tiles8, (size_t)n_tiles * 192));
  }
  if (c.n_phys > 1 || force_comm)
  {
    std::vector<int> ids(c.n_phys);
    for (int g = 0; g < G; g++)
      if (c.lead[g] == g)
        ids[c.comm_of[g]] = phys[g];
    ncclResult_t nr = ncclCommInitAll(c.comms.data(), c.n_phys, ids.data());
    if (nr != ncclSuccess)
    {
      int code = fail(RT_HIP_ERUNTIME, "ncclCommInitAll: %s", ncclGetErrorString(nr));
      ctx_release(c);
      (void)hipSetDevice(prev);
      return code;
    }
  }
  c.G = G;
  c.W = W;
  c.H = H;
  c.force_comm = force_comm;
  scene_serialise(c.scene_bytes, spheres, n_spheres, meshes, n_meshes);
  return RT_HIP_OK;
}

int render_image_impl(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes, size_t n_meshes,
                      const RtHipCamera *camera, const RtHipParams *params, int n_devices, float *h_image_rgb,
                      uint8_t *h_image_rgb8, uint64_t *h_stats, double *kernel_seconds)
{
  int rc = check_params(params);
  if (rc)
    return rc;
  if (!camera)
    return fail(RT_HIP_EINVAL, "camera is NULL");
  const int have = usable_devices();
  if (have < 1)
    return fail(RT_HIP_ENODEV, "no HIP device is available (this library has no CPU path)");
  const int W = params->width, H = params->height;
  const size_t n_px = (size_t)W * H;

  std::lock_guard<std::mutex> lock(g_ctx_mutex); /* one frame at a time: the context is shared */
  device_map_from_env();
  const int limit = g_device_map.empty() ? have : (int)g_device_map.size();
  if (n_devices < 1 || n_devices > limit)
    return fail(RT_HIP_ENODEV, "asked for %d devices, %d available%s", n_devices, limit, g_device_map.empty() ? "" : " in the device map");
  const int G = n_devices;
  std::vector<int> phys(G);
  for (int g = 0; g < G; g++)
  {
    phys[g] = g_device_map.empty() ? g : g_device_map[g];
    if (phys[g] < 0 || phys[g] >= have)
      return fail(RT_HIP_ENODEV, "device map entry %d -> %d: %d devices available", g, phys[g], have);
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  const auto tick0 = std::chrono::steady_clock::now();
  rc = ctx_prepare(spheres, n_spheres, meshes, n_meshes, G, phys, W, H, prev);
  if (rc)
    return rc;
  const auto tick1 = std::chrono::steady_clock::now();
  ImageCtx &c = g_ctx;
  std::vector<ImageCtx::Dev> &dev = c.dev;
  /* a failure below leaves the cached context in an unknown state: drop it */
#define IMG_TRY(expr)                                                                               \
  do                                                                                                \
  {                                                                                                 \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
    {                                                                                               \
      int c_ = fail(e_ == hipErrorOutOfMemory ? RT_HIP_ENOMEM : RT_HIP_ERUNTIME, "%s: %s", #expr,   \
                    hipGetErrorString(e_));                                                         \
      ctx_release(g_ctx);                                                                           \
      (void)hipSetDevice(prev);                                                                     \
      return c_;                                                                                    \
    }                                                                                               \
  } while (0)

  /* ---- launch every device's share ---- */
  for (int g = 0; g < G; g++)
  {
    ImageCtx::Dev &d = dev[g];
    IMG_TRY(hipSetDevice(phys[g]));
    const size_t slots = d.count ? d.count : 1;
    IMG_TRY(hipMemsetAsync(d.stats, 0, RT_HIP_NSTATS * sizeof(uint64_t), d.stream));
    IMG_TRY(hipMemsetAsync(d.tiles, 0, slots * 192 * sizeof(float), d.stream));  /* unrendered tiles stay black, */
    IMG_TRY(hipMemsetAsync(d.tiles8, 0, slots * 192, d.stream));                 /* like the reference's memset  */
    IMG_TRY(hipEventRecord(d.t0, d.stream));
  }
  /* Long frames are rendered in slabs (contiguous runs of each device's tile list) so that a
   * cancel request -- the CLI's SIGINT -- is honoured between slabs; what was finished is
   * still gathered and returned (the reference dumps its partial framebuffer on SIGINT,
   * main.c:37-48, from inside the signal handler; this does it from normal context). */
  const double work = (double)W * H * (double)params->samples;
  const uint32_t n_slabs = g_cancel ? (work > 4e9 ? 16u : (work > 2e8 ? 4u : 1u)) : 1u;
  bool cancelled = false;
  for (uint32_t slab = 0; slab < n_slabs && !cancelled; slab++)
  {
    for (int g = 0; g < G; g++)
    {
      ImageCtx::Dev &d = dev[g];
      const uint32_t k0 = (uint32_t)(((uint64_t)d.count * slab) / n_slabs);
      const uint32_t k1 = (uint32_t)(((uint64_t)d.count * (slab + 1)) / n_slabs);
      if (k1 == k0)
        continue;
      IMG_TRY(hipSetDevice(phys[g]));
      RtHipParams p = *params;
      p.tile_first = (uint32_t)g + k0 * (uint32_t)G;
      p.tile_stride = (uint32_t)G;
      p.tile_count = k1 - k0;
      const uint32_t chunks = p.integrator == RT_HIP_CAST_RAY ? 1u : rt_hip_suggest_chunks_depth(d.scene, p.tile_count, p.samples, p.max_depth);
      if (chunks > 1 && !d.ws)
        IMG_TRY(hipMalloc(&d.ws, rt_hip_scene_chunk_workspace_bytes(d.scene, d.count)));
      rc = rt_hip_render_tiles_chunked(d.scene, camera, &p, chunks, d.ws, d.tiles + (size_t)k0 * 192,
                                       d.tiles8 + (size_t)k0 * 192, d.stats, d.stream);
      if (rc)
      {
        ctx_release(g_ctx);
        (void)hipSetDevice(prev);
        return rc;
      }
    }
    if (n_slabs > 1)
    {
      for (int g = 0; g < G; g++)
      {
        IMG_TRY(hipSetDevice(phys[g]));
        IMG_TRY(hipStreamSynchronize(dev[g].stream));
      }
      cancelled = g_cancel && *g_cancel != 0;
    }
  }
  for (int g = 0; g < G; g++)
  {
    IMG_TRY(hipSetDevice(phys[g]));
    IMG_TRY(hipEventRecord(dev[g].t1, dev[g].stream));
    IMG_TRY(hipEventRecord(c.done[g], dev[g].stream));
  }

  /* ---- gather on logical device 0 (the root): every segment straight to it ----
   * A segment whose sender shares the root's physical device is a device-to-device copy on the root's stream, ordered after
   * the sender's `done` event; any other travels by grouped ncclSend / ncclRecv (point-to-point over xGMI: a gather to one
   * root uses the root's direct links concurrently, there is no ring).  A physical device is ONE RCCL rank however many
   * logical devices it carries: the rank's sends go on its lead's stream, which first waits for the other senders' `done`.
   * force_comm: every segment, the root's own included, goes through RCCL. */
  IMG_TRY(hipSetDevice(phys[0]));
  std::vector<size_t> first_slot(G, 0);
  for (int g = 1; g < G; g++)
    first_slot[g] = first_slot[g - 1] + dev[g - 1].count;
  auto by_rccl = [&](int g) { return c.force_comm || phys[g] != phys[0]; };
  bool any_rccl = false;
  for (int g = 0; g < G; g++)
  {
    if (!dev[g].count || (g == 0 && !c.force_comm))
      continue;
    if (by_rccl(g))
    {
      any_rccl = true;
      if (c.lead[g] != g)
      {
        IMG_TRY(hipSetDevice(phys[g]));
        IMG_TRY(hipStreamWaitEvent(dev[c.lead[g]].stream, c.done[g], 0));
      }
    }
    else
    {
      IMG_TRY(hipSetDevice(phys[0]));
      IMG_TRY(hipStreamWaitEvent(dev[0].stream, c.done[g], 0));
      const size_t nf = (size_t)dev[g].count * 192;
      IMG_TRY(hipMemcpyAsync(c.all_tiles + first_slot[g] * 192, dev[g].tiles, nf * sizeof(float), hipMemcpyDeviceToDevice, dev[0].stream));
      IMG_TRY(hipMemcpyAsync(c.all_tiles8 + first_slot[g] * 192, dev[g].tiles8, nf, hipMemcpyDeviceToDevice, dev[0].stream));
    }
  }
  if (any_rccl)
  {
    ncclResult_t nr = ncclGroupStart();
    for (int g = 0; g < G && nr == ncclSuccess; g++)
    {
      if (!dev[g].count || (g == 0 && !c.force_comm) || !by_rccl(g))
        continue;
      const size_t nf = (size_t)dev[g].count * 192;
      const int from = c.comm_of[g];
      hipStream_t send_stream = dev[c.lead[g]].stream;
      nr = ncclSend(dev[g].tiles, nf, ncclFloat, 0, c.comms[from], send_stream);
      if (nr == ncclSuccess) nr = ncclSend(dev[g].tiles8, nf, ncclUint8, 0, c.comms[from], send_stream);
      if (nr == ncclSuccess) nr = ncclRecv(c.all_tiles + first_slot[g] * 192, nf, ncclFloat, from, c.comms[0], dev[0].stream);
      if (nr == ncclSuccess) nr = ncclRecv(c.all_tiles8 + first_slot[g] * 192, nf, ncclUint8, from, c.comms[0], dev[0].stream);
    }
    ncclResult_t ne = ncclGroupEnd();
    if (nr == ncclSuccess)
      nr = ne;
    if (nr != ncclSuccess)
    {
      int code = fail(RT_HIP_ERUNTIME, "RCCL gather: %s", ncclGetErrorString(nr));
      ctx_release(g_ctx);
      (void)hipSetDevice(prev);
      return code;
    }
  }
  /* scatter each device's segment into the row-major image (root) */
  IMG_TRY(hipSetDevice(phys[0]));
  for (int g = 0; g < G; g++)
  {
    if (!dev[g].count)
      continue;
    const bool local = g == 0 && !c.force_comm; /* the root's own tiles never travel -- unless force_comm sent them through RCCL */
    const float *src = local ? dev[0].tiles : c.all_tiles + first_slot[g] * 192;
    const uint8_t *src8 = local ? dev[0].tiles8 : c.all_tiles8 + first_slot[g] * 192;
    rc = rt_hip_untile(src, src8, W, H, (uint32_t)g, (uint32_t)G, dev[g].count, c.image, c.image8, dev[0].stream);
    if (rc)
    {
      ctx_release(g_ctx);
      (void)hipSetDevice(prev);
      return rc;
    }
  }
  for (int g = 0; g < G; g++)
  {
    IMG_TRY(hipSetDevice(phys[g]));
    IMG_TRY(hipStreamSynchronize(dev[g].stream));
  }

  const auto tick2 = std::chrono::steady_clock::now();
  /* ---- did every workgroup find its pool slots?  (the status word of each physical device) ---- */
  uint32_t fail_flags = 0;
  for (int g = 0; g < G; g++)
    if (c.lead[g] == g)
    {
      IMG_TRY(hipSetDevice(phys[g]));
      uint32_t f = 0;
      rc = status_take(phys[g], &f);
      if (rc)
      {
        ctx_release(g_ctx);
        (void)hipSetDevice(prev);
        return rc;
      }
      fail_flags |= f;
    }

  /* ---- results ---- */
  IMG_TRY(hipSetDevice(phys[0]));
  if (h_image_rgb)
    IMG_TRY(hipMemcpy(h_image_rgb, c.image, n_px * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (h_image_rgb8)
    IMG_TRY(hipMemcpy(h_image_rgb8, c.image8, n_px * 3, hipMemcpyDeviceToHost));
  double worst = 0;
  uint64_t sums[RT_HIP_NSTATS] = {0, 0, 0, 0};
  for (int g = 0; g < G; g++)
  {
    IMG_TRY(hipSetDevice(phys[g]));
    float ms = 0;
    IMG_TRY(hipEventElapsedTime(&ms, dev[g].t0, dev[g].t1));
    if (ms * 1e-3 > worst)
      worst = ms * 1e-3;
    uint64_t st[RT_HIP_NSTATS];
    IMG_TRY(hipMemcpy(st, dev[g].stats, sizeof st, hipMemcpyDeviceToHost));
    for (int k = 0; k < RT_HIP_NSTATS; k++)
      sums[k] += st[k];
  }
  if (h_stats)
    memcpy(h_stats, sums, sizeof sums);
  if (kernel_seconds)
    *kernel_seconds = worst;
  (void)hipSetDevice(prev);
#undef IMG_TRY
  {
    const auto tick3 = std::chrono::steady_clock::now();
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    g_last_phases[0] = secs(tick0, tick1); /* context: scene compare, or upload + buffers + workspaces + communicators */
    g_last_phases[1] = secs(tick1, tick2); /* launches, kernels, gather, scatter -- until every stream is idle */
    g_last_phases[2] = secs(tick2, tick3); /* the frame, bytes and counters over PCIe */
  }
  if (fail_flags)
    return status_to_error(fail_flags); /* the buffers hold what was rendered; the unrendered tiles read NaN / 255 */
  if (cancelled)
    return fail(RT_HIP_ECANCELLED, "render cancelled: the image holds the tiles finished so far");
  return RT_HIP_OK;
}

int set_device_map_impl(const int *map, int n)
{
  if (n < 0 || n > 64 || (n > 0 && !map))
    return fail(RT_HIP_EINVAL, "device map: 0 <= n <= 64 entries");
  const int have = usable_devices();
  for (int k = 0; k < n; k++)
    if (map[k] < 0 || map[k] >= have)
      return fail(RT_HIP_ENODEV, "device map entry %d -> %d: %d devices available", k, map[k], have);
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  g_device_map_env_read = true; /* an explicit map (or its removal) overrides RT_HIP_DEVICE_MAP */
  g_device_map.assign(map, map + n);
  return RT_HIP_OK;
}

void last_phases_impl(double out[3])
{
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  memcpy(out, g_last_phases, sizeof g_last_phases);
}

void release_cache_impl()
{
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    ctx_release(g_ctx);
  }
  pend_pools_release();
}

uint64_t cache_builds_impl()
{
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  return g_ctx.builds;
}

} // namespace
