// Camera.cpp -- behaviour of 453-skeleton/Camera.cpp:8-95.
#include "Camera.h"

#include <algorithm>
#include <cmath>

using rtmath::vec3;

namespace {
const double kPi = 3.14159265358979323846;
const double kHalfPi = 1.57079632679489661923;
}

Camera::Camera(float t, float p, float r) : theta(t), phi(p), radius(r), target(0.0f) {}

vec3 Camera::getPos() const {
    const vec3 dir(std::cos(theta) * std::sin(phi), std::sin(theta), std::cos(theta) * std::cos(phi));
    return radius * dir + target;
}

rtmath::mat4 Camera::getView() const { return rtmath::lookAt(getPos(), target, vec3(0.0f, 1.0f, 0.0f)); }

rtmath::mat4 Camera::getProj(float aspect) const { return rtmath::perspective(rtmath::radians(45.0f), aspect, 0.1f, 5000.f); }

vec3 Camera::getLookDir() const { return rtmath::normalize(target - getPos()); }

vec3 Camera::getViewDir() const { return rtmath::normalize(target - getPos()); }

void Camera::incrementTheta(float dt) {
    const float next = theta + dt / 100.0f;          // stay strictly inside (-pi/2, pi/2): no flip over the pole
    if (next < kHalfPi && next > -kHalfPi) theta = next;
}

void Camera::incrementPhi(float dp) {
    phi -= dp / 100.0f;
    if (phi > 2.0 * kPi) phi -= 2.0 * kPi;
    else if (phi < 0.0f) phi += 2.0 * kPi;
}

void Camera::incrementR(float dr) { radius = std::max(MIN_RADIUS, radius - dr); }

void Camera::pan(float dx, float dy) {
    const vec3 look = getLookDir();
    const vec3 right = rtmath::normalize(rtmath::cross(look, vec3(0.f, 1.f, 0.f)));
    const vec3 up = rtmath::normalize(rtmath::cross(right, look));
    target += (-dx * right + dy * up) * (radius * 0.001f);   // pan speed scales with zoom
}
