// oracle/ref_shim/mapping_prefix.hpp — force-included (-include) in front of /root/reference/src/laserMapping.cpp.
// TEST INFRASTRUCTURE ONLY.  The node's process() is `while (1) { while (frames queued) {...}; sleep_for(2 ms); }`
// (reference src/laserMapping.cpp:231-233,889-892).  The driver feeds one frame, calls process() and needs it to come
// back once the queue is drained: with the standard headers already seen, `sleep_for(x)` is rewritten to
// `sleep_for(x); return` — the only textual use in the reference is the idle sleep at the bottom of that outer loop.
#pragma once
#include <chrono>
#include <condition_variable>
#include <future>
#include <iostream>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>
#define sleep_for(x) sleep_for(x); return
