// include/utils.h -- umbrella the reference's train.cpp includes first (reference: include/utils.h
// -> common.h).  Only what the GNN drivers need.
#pragma once
#include <omp.h>
#include <algorithm>
#include <iomanip>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <vector>
