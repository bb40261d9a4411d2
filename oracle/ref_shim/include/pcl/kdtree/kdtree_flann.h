// stand-in header (TEST INFRASTRUCTURE): see shim/pcl_shim.hpp
#pragma once
#include "shim/pcl_shim.hpp"
