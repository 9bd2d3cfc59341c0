/* The boundary from plain C: nothing but include/mi_depth.h and libmi_depth.so -- no Python, no torch, no HIP header.
 * Mirrors the reference's README flow (README.md:18-40: device -> DepthPro::new / load -> infer_from_rgb -> depth, focal
 * length) and `bench/inference.rs:21-48` (zeros input). Build and run (tests/test_gpu_parity.py does both on the GPU box):
 *
 *   gcc -O2 -Iinclude examples/infer_c_abi.c -Lburn_depth_amd -lmi_depth -Wl,-rpath,$PWD/burn_depth_amd -lm -o /tmp/infer_c_abi
 *   /tmp/infer_c_abi [checkpoint [tiny]]
 *
 * `checkpoint` is what the reference's `DepthPro::load(&device, path)` takes (depth_pro/mod.rs:193-208): a Burn `.mpk` record
 * (NamedMpkFileRecorder<HalfPrecisionSettings>), read natively by the library, or the engine's safetensors container. A second
 * argument `tiny` loads it with the reduced `tiny16_128` configuration (`DepthPro::load_with_config`, mod.rs:200-208).
 * Without a weight file it creates the reduced `tiny16_128` configuration with the seeded synthetic weights the parity
 * tests use and prints values the Python mirror must reproduce bit for bit (same library, same seed). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_depth.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != 0) {                                                              \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, md_last_error());      \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char** argv) {
  md_device_t dev = NULL;
  md_model_t model = NULL;
  CHECK(md_device_open(0, &dev));

  md_depth_pro_cfg cfg;
  md_depth_pro_cfg_default(&cfg);
  int S = 1536;
  const int tiny = argc <= 1 || (argc > 2 && strcmp(argv[2], "tiny") == 0);
  if (tiny) { /* the reduced configuration (the CI-size preset of the tests) */
    cfg.patch_encoder_preset = "tiny16_128";
    cfg.image_encoder_preset = "tiny16_128";
    cfg.fov_encoder_preset = "tiny16_128";
    cfg.decoder_features = 64;
    cfg.precision = MD_PREC_F32;
    cfg.max_batch = 1;
    S = 512;
  }
  if (argc > 1 && tiny) { /* DepthPro::load_with_config(&device, path, cfg): a Burn .mpk record or a safetensors container */
    CHECK(md_depth_pro_load_with_config(dev, &cfg, argv[1], &model));
  } else if (argc > 1) { /* DepthPro::load(&device, path): the default configuration */
    CHECK(md_depth_pro_load(dev, argv[1], &model));
  } else { /* DepthPro::new(&device, cfg), seeded */
    CHECK(md_depth_pro_create(dev, &cfg, 0, MD_INIT_PARITY, &model));
  }

  /* infer_from_rgb on a synthetic gradient image (packed RGB bytes, src/inference.rs:128-137) */
  const int w = 96, h = 64;
  uint8_t* rgb = (uint8_t*)malloc((size_t)w * h * 3);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      rgb[(y * w + x) * 3 + 0] = (uint8_t)(x * 255 / (w - 1));
      rgb[(y * w + x) * 3 + 1] = (uint8_t)(y * 255 / (h - 1));
      rgb[(y * w + x) * 3 + 2] = (uint8_t)((x + y) & 255);
    }
  float* depth = (float*)malloc((size_t)w * h * sizeof(float));
  float focal = 0.f, fovy = 0.f;
  CHECK(md_infer_from_rgb(model, rgb, (size_t)w * h * 3, w, h, MD_MEM_HOST, depth, &focal, &fovy, MD_MEM_HOST, NULL));
  double sum = 0.0;
  int finite = 1;
  for (int i = 0; i < w * h; ++i) {
    sum += depth[i];
    finite = finite && isfinite(depth[i]) && depth[i] > 0.f;
  }
  printf("rgb %dx%d: depth[0]=%.9g depth[last]=%.9g mean=%.9g focallength_px=%.9g fovy_rad=%.9g finite_positive=%d\n", w, h,
         depth[0], depth[w * h - 1], sum / (w * h), focal, fovy, finite);

  /* DepthPro::infer on zeros [1,3,S,S] (bench/inference.rs:21-48), host pointers in and out */
  float* x = (float*)calloc((size_t)3 * S * S, sizeof(float));
  float* d2 = (float*)malloc((size_t)S * S * sizeof(float));
  float f2 = 0.f, fovx = 0.f, fovy2 = 0.f;
  CHECK(md_depth_pro_infer(model, x, 1, S, S, MD_MEM_HOST, d2, &f2, &fovx, &fovy2, MD_MEM_HOST, NULL));
  printf("zeros %dx%d: depth[0]=%.9g focallength_px=%.9g fovx_deg=%.9g\n", S, S, d2[0], f2, fovx);

  /* DepthPro::decoder_from_features / head_debug (depth_pro/mod.rs:262-307): the decoder and the depth head alone on the CALLER's tensors.
   * Level shapes come from the model; zero features exercise the plumbing (the biases still reach the outputs). */
  int64_t levels = 0, F = 0;
  CHECK(md_model_query(model, "decoder_levels", &levels));
  CHECK(md_model_query(model, "decoder_features", &F));
  md_nchw_view views[8];
  float* fus[8] = {0};
  int64_t s0 = 0, s_last = 0;
  for (int l = 0; l < (int)levels && l < 8; ++l) {
    char key[64];
    int64_t c = 0, sz = 0;
    snprintf(key, sizeof key, "decoder_level%d_channels", l);
    CHECK(md_model_query(model, key, &c));
    snprintf(key, sizeof key, "decoder_level%d_size", l);
    CHECK(md_model_query(model, key, &sz));
    views[l].data = (const float*)calloc((size_t)(c * sz * sz), sizeof(float));
    views[l].channels = (int)c; views[l].height = (int)sz; views[l].width = (int)sz;
    const int64_t e = l == 0 ? sz : 2 * sz;
    fus[l] = (float*)malloc((size_t)(F * e * e) * sizeof(float));
    if (l == 0) s0 = sz;
    s_last = sz;
  }
  float* dfeat = (float*)malloc((size_t)(F * s0 * s0) * sizeof(float));
  float* dlow = (float*)malloc((size_t)(F * s_last * s_last) * sizeof(float));
  CHECK(md_depth_pro_decoder_from_features(model, views, (int)levels, 1, MD_MEM_HOST, dfeat, dlow, fus, MD_MEM_HOST, NULL));
  md_nchw_view fv = {dfeat, (int)F, (int)s0, (int)s0};
  md_head_debug hd = {0};
  float* canon = (float*)malloc((size_t)(4 * s0 * s0) * sizeof(float));
  hd.canonical = canon;  /* the other five HeadDebug tensors are skipped (NULL) */
  CHECK(md_depth_pro_head_debug(model, &fv, 1, MD_MEM_HOST, &hd, MD_MEM_HOST, NULL));
  printf("replay: decoder feature[0]=%.9g fusion_0 == feature: %d, head canonical[0]=%.9g\n", dfeat[0],
         memcmp(dfeat, fus[0], (size_t)(F * s0 * s0) * sizeof(float)) == 0, canon[0]);
  const int rc_levels = md_depth_pro_decoder_from_features(model, views, (int)levels - 1, 1, MD_MEM_HOST, dfeat, NULL, NULL, MD_MEM_HOST, NULL);
  printf("four levels: status %d (%s)\n", rc_levels, rc_levels == MD_ERR_LEVELS ? "MD_ERR_LEVELS" : "unexpected");
  for (int l = 0; l < (int)levels && l < 8; ++l) { free((void*)views[l].data); free(fus[l]); }
  free(dfeat); free(dlow); free(canon);

  /* the reference's Err(String) on a wrong buffer length (src/inference.rs:90-95) is a status code here */
  const int rc = md_infer_from_rgb(model, rgb, 10, w, h, MD_MEM_HOST, depth, &focal, &fovy, MD_MEM_HOST, NULL);
  printf("short rgb buffer: status %d (%s)\n", rc, rc == MD_ERR_SHAPE ? "MD_ERR_SHAPE" : "unexpected");

  free(x); free(d2); free(depth); free(rgb);
  CHECK(md_model_destroy(model));
  CHECK(md_device_close(dev));
  return finite && rc == MD_ERR_SHAPE && rc_levels == MD_ERR_LEVELS ? 0 : 2;
}
