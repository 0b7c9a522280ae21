// Camera.h -- orbit camera with the reference's interface (453-skeleton/Camera.h:5-44);
// rtmath::mat4 / vec3 stand where glm::mat4 / vec3 stood (identical memory layout).
#pragma once

#include "rtmath.h"

class Camera {
public:
    Camera(float t, float p, float r);

    rtmath::mat4 getView() const;            // lookAt(eye, target, +Y)
    rtmath::vec3 getPos() const;             // eye = radius * dir(theta, phi) + target
    rtmath::vec3 getLookDir() const;

    void incrementTheta(float dt);
    void incrementPhi(float dp);
    void incrementR(float dr);

    float getTheta() const { return theta; }
    float getPhi() const { return phi; }
    float getR() const { return radius; }
    rtmath::mat4 getProj(float aspect) const;
    rtmath::vec3 getViewDir() const;

    void pan(float dx, float dy);
    void setTarget(const rtmath::vec3& newTarget) { target = newTarget; }
    const rtmath::vec3& getTarget() const { return target; }

    float theta;
    float phi;
    float radius;
    rtmath::vec3 target;

    const float MIN_RADIUS = 0.1f;
    const float MAX_RADIUS = 1000.0f;
};
