// CPU-only sanitizer driver for the checkpoint readers of the C ABI (safetensors container + Burn `.mpk` record,
// csrc/md_weights.cpp): mutates a valid file (truncations, byte flips in the structure bytes and anywhere) and feeds every
// mutant to md::read_container, converting every tensor a surviving directory points at. Built by tests/test_mpk_c_reader.py
// with g++ -fsanitize=address,undefined (GPU sanitizers are not available on the pool; the readers are host code):
// any out-of-bounds read, overflow or leak in the parsers fails the test. No GPU, no HIP runtime call.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

#include "md_weights.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: fuzz_checkpoint_readers FILE ITERATIONS [SEED]\n");
    return 2;
  }
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<unsigned char> base((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (base.empty()) return 2;
  const int iters = atoi(argv[2]);
  std::mt19937_64 rng(argc > 3 ? (unsigned long long)atoll(argv[3]) : 1ull);
  const std::string tmp = std::string(argv[1]) + ".mut";
  long parsed = 0, rejected = 0, tensors = 0;
  {
    md::Container c;
    if (md::read_container(argv[1], &c) != 0 || c.tensors.empty()) {
      fprintf(stderr, "the unmodified file does not parse: %s\n", md::get_error());
      return 3;
    }
  }
  for (int it = 0; it < iters; ++it) {
    std::vector<unsigned char> b = base;
    const int mode = (int)(rng() % 5);
    if (mode == 0) {
      b.resize((size_t)(rng() % (b.size() + 1)));  // truncation anywhere
    } else if (mode == 1) {
      b.resize(std::min<size_t>(b.size(), (size_t)(rng() % 512)));  // truncation inside the header
    } else {
      const size_t lim = mode == 4 ? b.size() : std::min<size_t>(b.size(), 2048);  // mostly the structure bytes at the start
      const int n = 1 + (int)(rng() % 6);
      for (int i = 0; i < n; ++i) b[(size_t)(rng() % lim)] = (unsigned char)rng();
    }
    {
      std::ofstream o(tmp, std::ios::binary | std::ios::trunc);
      o.write((const char*)b.data(), (std::streamsize)b.size());
    }
    md::Container c;
    if (md::read_container(tmp.c_str(), &c) != 0) {
      ++rejected;
      continue;
    }
    ++parsed;
    for (auto& kv : c.tensors) {  // touch every payload the directory points at
      size_t n = 1;
      bool sane = true;
      for (int64_t d : kv.second.shape) {
        if (d < 0 || (d > 0 && n > (size_t)1 << 28)) { sane = false; break; }
        n *= (size_t)d;
      }
      if (!sane || n > (size_t)1 << 26) continue;
      std::vector<float> out(n);
      (void)md::container_tensor_to_f32(c, kv.second, out.data(), n);
      ++tensors;
    }
  }
  remove(tmp.c_str());
  printf("iterations %d: parsed %ld (tensors converted %ld), rejected %ld\n", iters, parsed, tensors, rejected);
  return 0;
}
