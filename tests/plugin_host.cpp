// Test harness: loads a model plugin exactly like the reference's plugin_manager
// (include/kmdiff/model_manager.hpp:33-94: dlopen RTLD_LAZY, "plugin_name", "create" + bits,
// configure(config), process(controls, cases)) and prints one line per row read from stdin:
//   p sign mean_control mean_case      (doubles as C99 hex floats)
// usage: plugin_host <plugin.so> <bits: 8|16|32> <config> <nc> <nk>  < rows.txt
#include <dlfcn.h>

#include <cstdio>
#include <iostream>
#include <memory>

#include "../kmdiff_amd/host/imodel_abi.hpp"

template <size_t MAX_C>
int run(void* h, const std::string& config, size_t nc, size_t nk, const char* sym)
{
  using model_t = kmdiff::IModel<MAX_C>;
  using count_t = typename model_t::count_type;
  auto create = reinterpret_cast<model_t* (*)()>(dlsym(h, sym));
  if (!create) { std::fprintf(stderr, "%s\n", dlerror()); return 2; }
  std::shared_ptr<model_t> model(create());              // host owns and deletes (model_manager.hpp:58-63)
  model->configure(config);
  std::vector<count_t> row(nc + nk);
  for (;;)
  {
    for (size_t i = 0; i < nc + nk; ++i)
    {
      unsigned long v;
      if (!(std::cin >> v)) return 0;
      row[i] = (count_t)v;
    }
    kmdiff::Range<count_t> controls(row, 0, nc), cases(row, nc, nk);
    auto [p, sign, mc, mk] = model->process(controls, cases);
    std::printf("%a %d %a %a\n", p, (int)sign, mc, mk);
  }
}

int main(int argc, char** argv)
{
  if (argc < 6) return 1;
  void* h = dlopen(argv[1], RTLD_LAZY);
  if (!h) { std::fprintf(stderr, "%s\n", dlerror()); return 2; }
  auto name = reinterpret_cast<std::string (*)()>(dlsym(h, "plugin_name"));
  if (!name) { std::fprintf(stderr, "%s\n", dlerror()); return 2; }
  std::fprintf(stderr, "plugin: %s\n", name().c_str());
  const std::string bits = argv[2], config = argv[3];
  const size_t nc = std::stoul(argv[4]), nk = std::stoul(argv[5]);
  try
  {
    if (bits == "8") return run<kmdiff::maxc8>(h, config, nc, nk, "create8");
    if (bits == "16") return run<kmdiff::maxc16>(h, config, nc, nk, "create16");
    return run<kmdiff::maxc32>(h, config, nc, nk, "create32");
  }
  catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 3; }
}
