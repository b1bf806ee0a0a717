/* main.c -- command-line host, the counterpart of the reference's main.c.
 *
 * Same flags (-w -h -s -o, reference main.c:149-174; note -h is HEIGHT there
 * too) and the same default scene (the 38-sphere room, main.c:244-397, camera
 * main.c:425, seed main.c:182), through the same API calls a reference caller
 * makes: init_camera(), render(), stbi_write_png().  Extra flags select what
 * the reference fixes at compile time:
 *   -d <depth>   bounce limit (reference MAX_DEPTH, raytracer.h:25; default 5)
 *   -c <config>  scene 1..5 of BASELINE.json (default 4, the reference's room)
 *   -g <gpus>    GPUs of this node to spread the image over (default 1)
 *   -r <seed>    RNG seed (default 1666943821)
 *   -i <0|1>     integrator: 0 trace_path (default), 1 cast_ray -- the `#if 1` of
 *                raytracer.c:207-211
 * Timing is wall-clock (the reference's clock()/integer division, main.c:427-433,
 * reports summed CPU time truncated to seconds -- deliberately not reproduced).
 * SIGINT: the reference's handler writes and frees the live framebuffer from
 * signal context (main.c:37-48); here it only sets a flag that the renderer
 * polls between slabs of tiles, and the partial image is written normally.
 */
#define _POSIX_C_SOURCE 200809L
#include <signal.h>
#include <time.h>

#include "raytracer.h"
#include "rt_hip.h"
#include "scenes.h"

int stbi_write_png(char const *filename, int w, int h, int comp, const void *data, int stride_in_bytes);

static volatile int interrupted = 0; /* polled by the renderer between slabs of tiles */
static void on_sigint(int sig)
{
  (void)sig;
  interrupted = 1;
}

static double now_seconds(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct
{
  Options options;
  int depth, config, gpus, integrator;
  uint64_t seed;
} Args;

static void usage(const char *prog)
{
  fprintf(stderr,
          "Usage: %s -w <width> -h <height> -s <samples per pixel> -o <filename>\n"
          "          [-d <max depth>] [-c <scene config 1..5>] [-g <gpus>] [-r <seed>]\n"
          "          [-i <integrator: 0 trace_path, 1 cast_ray>]\n",
          prog);
}

static int parse_args(int argc, char **argv, Args *a)
{
  for (int i = 1; i < argc; i++)
  {
    if (argv[i][0] != '-' || argv[i][1] == '\0' || i + 1 >= argc)
      return -1;
    const char *val = argv[++i];
    switch (argv[i - 1][1])
    {
    case 'w': a->options.width = atoi(val); break;
    case 'h': a->options.height = atoi(val); break;
    case 's': a->options.samples = atoi(val); break;
    case 'o': a->options.result = (char *)val; break;
    case 'd': a->depth = atoi(val); break;
    case 'c': a->config = atoi(val); break;
    case 'g': a->gpus = atoi(val); break;
    case 'r': a->seed = strtoull(val, NULL, 10); break;
    case 'i': a->integrator = atoi(val); break;
    default: return -1;
    }
  }
  return 0;
}

int main(int argc, char **argv)
{
  Args a;
  memset(&a, 0, sizeof a);
  a.options.width = 320; /* reference main.c:24-30 */
  a.options.height = 180;
  a.options.samples = 50;
  a.options.result = "result.png";
  a.options.obj = "assets/cube.obj";
  a.depth = MAX_DEPTH;
  a.config = 4;
  a.gpus = 1;
  a.seed = 1666943821ull;

  if (argc <= 1 || parse_args(argc, argv, &a) != 0)
  {
    usage(argv[0]);
    return EXIT_FAILURE;
  }
  RtSceneInfo info;
  if (rt_scene_info(a.config, &info) != 0 || a.options.width < 2 || a.options.height < 2 || a.options.samples < 1 ||
      (a.integrator != RT_TRACE_PATH && a.integrator != RT_CAST_RAY))
  {
    usage(argv[0]);
    return EXIT_FAILURE;
  }
  printf("seed = %llu\n", (unsigned long long)a.seed);

  Object *scene = (Object *)calloc(info.n_objects ? info.n_objects : 1, sizeof(Object));
  MeshObject *meshes = (MeshObject *)calloc(info.n_meshes ? info.n_meshes : 1, sizeof(MeshObject));
  size_t fb_len = (size_t)a.options.width * (size_t)a.options.height * 3;
  uint8_t *framebuffer = (uint8_t *)calloc(fb_len, 1);
  if (!scene || !meshes || !framebuffer ||
      rt_scene_build(a.config, a.options.width, a.options.height, scene, meshes) != 0)
  {
    fprintf(stderr, "could not allocate framebuffer or scene\n");
    return EXIT_FAILURE;
  }
  signal(SIGINT, on_sigint);
  rt_set_cancel_flag(&interrupted);

  Camera camera;
  vec3 pos = {info.cam_pos[0], info.cam_pos[1], info.cam_pos[2]};
  vec3 target = {info.cam_target[0], info.cam_target[1], info.cam_target[2]};
  init_camera(&camera, pos, target, &a.options);

  rt_set_max_depth(a.depth);
  rt_set_seed(a.seed);
  rt_set_devices(a.gpus);
  rt_set_integrator(a.integrator);

  /* the process's phase clock (bench.py's cli_host entry reads the `phases:` line): the HIP runtime comes up at the first
   * device query; render_ex's own split is the shim's (rt_hip_last_image_phases) */
  const double t_start = now_seconds();
  (void)rt_hip_device_count();
  const double t_hip = now_seconds();

  double tic = now_seconds();
  render_ex(framebuffer, NULL, scene, info.n_objects, meshes, info.n_meshes, &camera, &a.options);
  double toc = now_seconds();
  double phase[3] = {0, 0, 0};
  rt_hip_last_image_phases(phase);

  const double kernel_s = rt_last_render_seconds();
  printf("%d x %d (%d) pixels\n", a.options.width, a.options.height, a.options.width * a.options.height);
  printf("cast %lld rays\n", ray_count);
  printf("checked %lld possible intersections\n", intersection_test_count);
  printf("rendering took %f seconds (GPU kernels %f s on %d GPU%s)\n", toc - tic, kernel_s, a.gpus,
         a.gpus == 1 ? "" : "s");
  if (kernel_s > 0)
    printf("%.3e ray-bounces/s, %.2f Mpixel-samples/s\n", (double)rt_last_ray_bounces() / kernel_s,
           (double)a.options.width * a.options.height * a.options.samples / kernel_s * 1e-6);
  int status = EXIT_SUCCESS;
  if (rt_last_render_cancelled())
    printf("interrupted: the image holds the tiles finished so far\n");
  printf("writing result to '%s'...\n", a.options.result);
#ifndef VALGRIND
  if (stbi_write_png(a.options.result, a.options.width, a.options.height, 3, framebuffer, a.options.width * 3) == 0)
    status = EXIT_FAILURE;
  else
    printf("done.\n");
#endif
  printf("phases: HIP runtime start %.6f s, context %.6f s, render %.6f s, copy out %.6f s, PNG %.6f s\n", t_hip - t_start, phase[0],
         phase[1], phase[2], now_seconds() - toc);
  rt_scene_free_meshes(meshes, info.n_meshes);
  free(meshes);
  free(scene);
  free(framebuffer);
  return status;
}
