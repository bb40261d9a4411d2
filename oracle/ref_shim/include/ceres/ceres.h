// stand-in header (TEST INFRASTRUCTURE): see shim/ceres_shim.hpp
#pragma once
#include <algorithm>
#include "shim/ceres_shim.hpp"
