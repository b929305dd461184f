#pragma once
#include "cuda_runtime_api.h"
