// dataset root: $DATASET_PATH (must end with '/', as in the reference's include/gnn/configs.h:5);
// unlike the reference, an unset variable is reported instead of crashing at static-init time (Q15).
#pragma once
#include <cstdlib>
#include <iostream>
#include <string>

inline std::string dataset_root() {
  const char* p = std::getenv("DATASET_PATH");
  if (!p) {
    std::cerr << "DATASET_PATH is not set (directory holding <dataset>/graph.meta.txt, with a trailing '/')\n";
    exit(1);
  }
  std::string s(p);
  if (!s.empty() && s.back() != '/') s += '/';
  return s;
}
