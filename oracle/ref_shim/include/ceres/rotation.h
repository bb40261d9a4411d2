// stand-in for <ceres/rotation.h>: included by reference src/lidarFactor.hpp:5, nothing from it is used
#pragma once
