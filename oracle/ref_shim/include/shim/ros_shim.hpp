// oracle/ref_shim/include/shim/ros_shim.hpp — stand-in for the slice of roscpp / ROS messages / tf the A-LOAM nodes touch.
//
// TEST INFRASTRUCTURE ONLY.  Purpose: let the reference's own translation units (src/scanRegistration.cpp,
// src/laserOdometry.cpp, src/laserMapping.cpp under /root/reference, compiled where they lie, never copied) build and
// run without ROS, so that the oracle's restatement of THEIR code can be checked against the code itself.
// Messages are plain structs; "publishing" appends a copy to a per-topic list the driver reads back; "subscribing"
// records the callback so the driver (or ros::spinOnce through a hook) can deliver messages in a scripted order.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace boost { using std::shared_ptr; }   // the message typedefs below are boost::shared_ptr in ROS-1

namespace ros {
struct Time {
  double sec = 0.0;
  Time() {}
  Time& fromSec(double s) { sec = s; return *this; }
  double toSec() const { return sec; }
  static Time now() { return Time(); }
  bool operator==(const Time& o) const { return sec == o.sec; }
  bool operator<(const Time& o) const { return sec < o.sec; }
};
struct Duration { double sec; explicit Duration(double s = 0) : sec(s) {} void sleep() const {} };
struct Rate { explicit Rate(double) {} void sleep() {} };
}  // namespace ros

namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }

namespace sensor_msgs {
struct PointField { enum { INT8 = 1, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 }; std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0; };
struct PointCloud2 {
  std_msgs::Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
  typedef std::shared_ptr<PointCloud2> Ptr;
  typedef std::shared_ptr<const PointCloud2> ConstPtr;
};
typedef PointCloud2::Ptr PointCloud2Ptr;
typedef PointCloud2::ConstPtr PointCloud2ConstPtr;
struct Imu { std_msgs::Header header; typedef std::shared_ptr<const Imu> ConstPtr; };
}  // namespace sensor_msgs

namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[36] = {0}; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
}  // namespace geometry_msgs

namespace nav_msgs {
struct Odometry {
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
  typedef std::shared_ptr<Odometry> Ptr;
  typedef std::shared_ptr<const Odometry> ConstPtr;
};
struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
}  // namespace nav_msgs

namespace ref_shim {
// per message type: what was published on each topic, and who subscribed to each topic
template <class M> std::map<std::string, std::vector<M>>& published() { static std::map<std::string, std::vector<M>> m; return m; }
template <class M> std::map<std::string, std::function<void(const std::shared_ptr<const M>&)>>& subscribers() {
  static std::map<std::string, std::function<void(const std::shared_ptr<const M>&)>> m; return m;
}
inline std::map<std::string, double>& params() { static std::map<std::string, double> m; return m; }
inline std::function<void()>& spin_hook() { static std::function<void()> f; return f; }
inline std::function<bool()>& ok_hook() { static std::function<bool()> f; return f; }
template <class M> void deliver(const std::string& topic, const M& msg) {
  auto it = subscribers<M>().find(topic);
  if (it == subscribers<M>().end()) { std::fprintf(stderr, "ref_shim: nobody subscribed to %s\n", topic.c_str()); std::abort(); }
  it->second(std::make_shared<const M>(msg));
}
}  // namespace ref_shim

namespace ros {
inline void init(int&, char**, const std::string&) {}
inline bool ok() { return ref_shim::ok_hook() ? ref_shim::ok_hook()() : false; }
inline void spinOnce() { if (ref_shim::spin_hook()) ref_shim::spin_hook()(); }
inline void spin() { while (ok()) spinOnce(); }

struct Publisher {
  std::string topic;
  template <class M> void publish(const M& m) const { ref_shim::published<M>()[topic].push_back(m); }
};
struct Subscriber { std::string topic; };

struct NodeHandle {
  template <class T> bool param(const std::string& name, T& var, const T& def) const {
    auto it = ref_shim::params().find(name);
    if (it == ref_shim::params().end()) { var = def; return false; }
    var = static_cast<T>(it->second);
    return true;
  }
  template <class M> Subscriber subscribe(const std::string& topic, uint32_t, void (*cb)(const std::shared_ptr<const M>&)) {
    ref_shim::subscribers<M>()[topic] = cb;
    return Subscriber{topic};
  }
  template <class M> Publisher advertise(const std::string& topic, uint32_t) { ref_shim::published<M>()[topic]; return Publisher{topic}; }
};
}  // namespace ros

#define ROS_INFO(...) do { std::printf(__VA_ARGS__); std::printf("\n"); } while (0)
#define ROS_WARN(...) do { std::printf(__VA_ARGS__); std::printf("\n"); } while (0)
#define ROS_ERROR(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_BREAK() do { std::fprintf(stderr, "ROS_BREAK at %s:%d\n", __FILE__, __LINE__); std::abort(); } while (0)

namespace tf {
struct Vector3 { double x, y, z; Vector3(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {} };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; void setX(double v) { x = v; } void setY(double v) { y = v; } void setZ(double v) { z = v; } void setW(double v) { w = v; } };
struct Transform { Vector3 origin; Quaternion rotation; void setOrigin(const Vector3& o) { origin = o; } void setRotation(const Quaternion& q) { rotation = q; } };
struct StampedTransform : Transform {
  ros::Time stamp; std::string frame_id, child_frame_id;
  StampedTransform(const Transform& t, const ros::Time& s, const std::string& f, const std::string& c) : Transform(t), stamp(s), frame_id(f), child_frame_id(c) {}
};
struct TransformBroadcaster { std::vector<StampedTransform> sent; void sendTransform(const StampedTransform& t) { sent.push_back(t); } };
}  // namespace tf
