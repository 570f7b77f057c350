// Command-line front-end: `./raytracer [-t numThreads] filepath`, the interface of the reference's src/main.cpp:8-37,
// with the render loop running on MI355X GPUs through the C ABI of include/rt_amd.h.
//
// Kept from the reference: the argument forms (-tN attached or detached), the stdout lines ("Scanned ...", "BVH created
// with ...", "Default render." / "Fisheye enabled." / "Depth of Field enabled.", the phase timings printed at exit),
// the output file named by the scene, exit codes 1 (usage / unreadable scene) and -1 (unknown flag).
// Added (none of them changes `./raytracer [-tN] file`):
//   --gpus N        split the frame by interleaved image tiles over N devices of this node, gather on device 0
//   --share-gpus    with --gpus N on a node with fewer than N devices: several parts per device (a rehearsal of the
//                   N-way split and gather, not a speed-up)
//   --tile WxH      tile size of that split (default 32x32)
//   --spp-pass N    camera samples per pixel traced per wavefront pass
//   --seed N        seed of the counter-based random stream (default 56, the reference's unused `seed` default)
//   --dump-hits F   write primary-ray hit ids and t (int32, float32 per pixel, row-major) to F and exit
//   --stats         print ray counts, Mrays/s and per-kernel milliseconds
//   --adaptive T    adaptive sampling: a pixel stops taking samples once the standard error of its mean luminance is at
//                   most T * (mean + 0.05), checked every --adaptive-batch samples (default 16)
//   --device-bvh    skip the host BVH build; the GPU builds a linear BVH of its own (milliseconds instead of seconds for
//                   millions of objects; not the reference's tree: objects hit at exactly equal t may swap)
// `-t` keeps meaning host threads: it sets how many the parsers and the BVH build use (default: all, at most 16).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_host.h"

namespace {

struct Phase {
  const char *name;
  double seconds;
};

void printPhase(const Phase &p) {  // Stats::print, src/acceleration/Profiler.cpp:13-19
  int minutes = (int)(p.seconds / 60);
  double seconds = p.seconds - minutes * 60;
  std::printf("%s: %d min %.2f sec\n", p.name, minutes, seconds);
}

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int usage(const char *argv0, int code) {
  std::fprintf(stderr, "usage: %s [-t numThreads] filepath\n", argv0);
  std::fprintf(stderr, "       further options: --gpus N [--share-gpus] --stats --seed S --tile WxH --spp-pass K --device-bvh\n"
                       "                        --adaptive THRESHOLD [--adaptive-batch B] --dump-hits FILE\n");
  return code;
}

}  // namespace

int main(int argc, char **argv) {
  int numThreads = 0;  // 0 = not given: the host library picks (the reference's default of 4 is a CPU-render setting)
  int gpus = 1, tileW = 32, tileH = 32, sppPass = 0;
  unsigned seed = 56;
  bool stats = false, shareGpus = false, deviceBvh = false;
  float adaptive = 0.0f;
  int adaptiveBatch = 0;
  std::string dumpHits, path;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&](const char *what) -> const char * {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(1); }
      return argv[++i];
    };
    if (a.rfind("-t", 0) == 0 && a.size() > 2 && a[2] != '-') numThreads = std::atoi(a.c_str() + 2);
    else if (a == "-t") numThreads = std::atoi(next("-t"));
    else if (a == "--gpus") gpus = std::atoi(next("--gpus"));
    else if (a == "--tile") { if (std::sscanf(next("--tile"), "%dx%d", &tileW, &tileH) != 2) return usage(argv[0], 1); }
    else if (a == "--spp-pass") sppPass = std::atoi(next("--spp-pass"));
    else if (a == "--seed") seed = (unsigned)std::strtoul(next("--seed"), nullptr, 10);
    else if (a == "--dump-hits") dumpHits = next("--dump-hits");
    else if (a == "--stats") stats = true;
    else if (a == "--share-gpus") shareGpus = true;
    else if (a == "--device-bvh") deviceBvh = true;
    else if (a == "--adaptive") adaptive = (float)std::atof(next("--adaptive"));
    else if (a == "--adaptive-batch") adaptiveBatch = std::atoi(next("--adaptive-batch"));
    else if (!a.empty() && a[0] == '-') return usage(argv[0], -1);
    else path = a;
  }
  if (path.empty()) return usage(argv[0], 1);
  if (numThreads > 0) rth_set_threads(numThreads);  // host threads: parsing, triangle set-up, BVH build
  if (gpus < 1) gpus = 1;

  // The HIP runtime takes a few hundred milliseconds to come up: let it, while the host reads the scene.
  std::thread warmUp([] { (void)rt_device_count(); });
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joinWarmUp{warmUp};

  std::vector<Phase> phases;
  double t0 = now();
  RthScene *hscene = rth_scene_load(path.c_str());
  if (!hscene) {
    std::fprintf(stderr, "%s\n", rth_last_error());
    return 1;
  }
  int printed = 0;
  auto flushLog = [&]() {
    for (; printed < rth_scene_log_count(hscene); ++printed) std::printf("%s\n", rth_scene_log_line(hscene, printed));
  };
  flushLog();
  phases.push_back({"Scene construction", now() - t0});

  t0 = now();
  if ((deviceBvh ? rth_scene_flatten_without_bvh(hscene) : rth_scene_build(hscene)) != RT_OK) {
    std::fprintf(stderr, "%s\n", rth_last_error());
    return 1;
  }
  flushLog();
  phases.push_back({"BVH construction", now() - t0});
  const RtSceneDesc *desc = rth_scene_desc(hscene);
  if (desc->options.fisheye) std::printf("Fisheye enabled.\n");
  else if (desc->options.focus > 0) std::printf("Depth of Field enabled.\n");
  else std::printf("Default render.\n");

  const int W = desc->options.width, H = desc->options.height;
  std::vector<RtScene *> scenes(gpus, nullptr);
  const double tDevice = now();
  warmUp.join();
  const int devices = rt_device_count();
  if (gpus > devices && !(shareGpus && devices > 0)) {
    std::fprintf(stderr, "--gpus %d, but this node offers %d device(s)\n", gpus, devices);
    return 1;
  }
  for (int g = 0; g < gpus; ++g) {
    scenes[g] = rt_scene_create(desc, g % devices);
    if (!scenes[g]) {
      std::fprintf(stderr, "device %d: %s\n", g, rt_last_error());
      return 1;
    }
  }
  const double sDevice = now() - tDevice;
  if (deviceBvh) std::printf("BVH built on the GPU in %.2f ms (linear BVH, not the reference's tree)\n", rt_scene_bvh_build_ms(scenes[0]));

  if (!dumpHits.empty()) {
    std::vector<int32_t> ids((size_t)W * H);
    std::vector<float> t((size_t)W * H);
    if (rt_trace_primary(scenes[0], 0, 0, W, H, ids.data(), t.data()) != RT_OK) {
      std::fprintf(stderr, "%s\n", rt_last_error());
      return 1;
    }
    FILE *f = std::fopen(dumpHits.c_str(), "wb");
    if (!f) { std::perror(dumpHits.c_str()); return 1; }
    for (size_t i = 0; i < ids.size(); ++i) { std::fwrite(&ids[i], 4, 1, f); std::fwrite(&t[i], 4, 1, f); }
    std::fclose(f);
    return 0;
  }

  t0 = now();
  std::vector<float> frame((size_t)W * H * 4, 0.0f);
  std::vector<RtStats> st(gpus);
  std::vector<int> rc(gpus, RT_OK);
  std::vector<std::string> errs(gpus);
  if (gpus == 1) {
    RtRenderParams p{};
    p.rngMode = RT_RNG_COUNTER; p.seed = seed; p.partCount = 1; p.tileW = tileW; p.tileH = tileH; p.samplesPerPass = sppPass;
    p.adaptiveThreshold = adaptive; p.adaptiveBatch = adaptiveBatch;
    rc[0] = rt_render(scenes[0], &p, frame.data(), &st[0]);
    if (rc[0] != RT_OK) errs[0] = rt_last_error();
  } else {
    // One host thread per device; every device renders its interleaved tiles of the frame and keeps them in its own
    // memory. Device 0 then collects the partial frames device to device (pixels outside a device's tiles are 0, so
    // adding is exact) and the host reads the finished frame once.
    std::vector<std::thread> workers;
    for (int g = 0; g < gpus; ++g) {
      workers.emplace_back([&, g]() {
        RtRenderParams p{};
        p.rngMode = RT_RNG_COUNTER; p.seed = seed; p.partIndex = g; p.partCount = gpus; p.tileW = tileW; p.tileH = tileH;
        p.samplesPerPass = sppPass;
        p.adaptiveThreshold = adaptive; p.adaptiveBatch = adaptiveBatch;
        rc[g] = rt_render(scenes[g], &p, nullptr, &st[g]);
        if (rc[g] != RT_OK) errs[g] = rt_last_error();
      });
    }
    for (auto &w : workers) w.join();
    bool allOk = true;
    for (int g = 0; g < gpus; ++g) allOk = allOk && rc[g] == RT_OK;
    if (allOk) {
      // one gather: every other device's own pixels (1/N of the frame each) straight to device 0, all copies in flight at once
      int grc = rt_framebuffer_gather(scenes[0], scenes.data() + 1, gpus - 1);
      if (grc != RT_OK) { rc[0] = grc; errs[0] = rt_last_error(); }
    }
    if (rc[0] == RT_OK) {
      rc[0] = rt_framebuffer_read(scenes[0], frame.data());
      if (rc[0] != RT_OK) errs[0] = rt_last_error();
    }
  }
  for (int g = 0; g < gpus; ++g)
    if (rc[g] != RT_OK) {
      std::fprintf(stderr, "device %d: %s\n", g, errs[g].c_str());
      return 1;
    }
  phases.push_back({"Rendering", now() - t0});

  const double tPng = now();
  std::vector<uint8_t> rgba8((size_t)W * H * 4);
  rth_to_srgb8(frame.data(), (int64_t)W * H, rgba8.data());
  if (rth_png_write_rgba8(rth_scene_filename(hscene), W, H, rgba8.data()) != RT_OK)
    std::fprintf(stderr, "PNG encoding error: %s\n", rth_last_error());  // the reference reports and carries on

  if (stats) {
    std::printf("Device scene (runtime start, upload): %.2f s; PNG: %.2f s\n", sDevice, now() - tPng);
    unsigned long long rays = 0, samples = 0;
    double msMax = 0;
    for (int g = 0; g < gpus; ++g) {
      rays += st[g].raysClosest + st[g].raysAny;
      samples += st[g].cameraSamples;
      msMax = std::max(msMax, st[g].msTotal);
      std::printf("GPU %d: %llu rays, %llu camera samples, %.2f ms (extend %.2f, shade %.2f, shadow %.2f)\n", g,
                  (unsigned long long)(st[g].raysClosest + st[g].raysAny), (unsigned long long)st[g].cameraSamples,
                  st[g].msTotal, st[g].msExtend, st[g].msShade, st[g].msShadow);
    }
    if (msMax > 0)
      std::printf("Total: %llu rays, %.1f Mrays/s, %.1f Msamples/s on %d GPU(s)\n", rays, rays / msMax / 1e3, samples / msMax / 1e3, gpus);
  }
  for (const Phase &p : phases) printPhase(p);
  for (RtScene *s : scenes) rt_scene_destroy(s);
  rth_scene_free(hscene);
  return 0;
}
