// see include/layers/output_layers.h (all head layers of the trainer are declared there)
#pragma once
#include "output_layers.h"
