// Test helper (tests/test_checkpoint_pt.py): a libtorch module that registers its parameters exactly the way the reference's
// LocalMap does (include/neural_net/local_map.cpp:29-55,73-75) and is saved / loaded with torch::save / torch::load like
// NeuralSLAM::export_checkpoint / load_checkpoint (include/neural_mapping/neural_mapping.cpp:1331-1351).
//   pt_roundtrip save <impl 0|1> <table_floats> <path>   parameters filled with a fixed pattern, torch::save
//   pt_roundtrip load <impl 0|1> <table_floats> <path>   torch::load, prints "name numel sum sum_of_squares" per parameter
#include <torch/torch.h>

#include <iomanip>
#include <iostream>

struct MiniLocalMap : torch::nn::Module {
  torch::Tensor table, flat_decoder;
  torch::nn::Sequential decoder;
  MiniLocalMap(int impl, int64_t n_table) {
    table = register_parameter("encoder_local_map", torch::zeros({n_table}), true);
    if (impl == 0) {
      decoder->push_back(torch::nn::Linear(32, 64));
      decoder->push_back(torch::nn::ReLU(true));
      for (int i = 0; i < 3; ++i) {
        decoder->push_back(torch::nn::Linear(64, 64));
        decoder->push_back(torch::nn::ReLU(true));
      }
      decoder->push_back(torch::nn::Linear(64, 2));
      decoder = register_module("decoder", decoder);
    } else {
      flat_decoder = register_parameter("decoder", torch::zeros({32 * 64 + 2 * 64 * 64 + 64 * 2}), true);
    }
  }
};

int main(int argc, char **argv) {
  if (argc != 5) return 2;
  const std::string mode = argv[1];
  const int impl = std::atoi(argv[2]);
  const int64_t n_table = std::atoll(argv[3]);
  auto m = std::make_shared<MiniLocalMap>(impl, n_table);
  torch::NoGradGuard ng;
  if (mode == "save") {
    int k = 0;
    for (auto &p : m->named_parameters()) {
      p.value().copy_(torch::sin(torch::arange(p.value().numel(), torch::kFloat32) * 0.37f + (float)k).view_as(p.value()));
      ++k;
    }
    torch::save(m, argv[4]);
  } else {
    torch::load(m, argv[4]);
  }
  std::cout << std::setprecision(9);
  for (auto &p : m->named_parameters()) {
    auto v = p.value().to(torch::kFloat64);
    std::cout << p.key() << " " << v.numel() << " " << v.sum().item<double>() << " " << (v * v).sum().item<double>() << "\n";
  }
  return 0;
}
