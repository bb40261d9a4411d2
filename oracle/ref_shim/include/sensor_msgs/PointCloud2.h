// stand-in header (TEST INFRASTRUCTURE): see shim/ros_shim.hpp
#pragma once
#include "shim/ros_shim.hpp"
