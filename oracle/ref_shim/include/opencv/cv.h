// stand-in for <opencv/cv.h>: the reference includes it (src/scanRegistration.cpp:44) but uses nothing from it
#pragma once
