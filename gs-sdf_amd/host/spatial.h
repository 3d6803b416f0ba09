// spatial.h — drop-in for simple-knn's header (neural_gaussian.cpp:14); call site neural_gaussian.cpp:314.
#pragma once
#include <torch/torch.h>

// mean of the squared distances to the 3 nearest neighbours of every point, [N,3] -> [N]
torch::Tensor distCUDA2(const torch::Tensor &points);
