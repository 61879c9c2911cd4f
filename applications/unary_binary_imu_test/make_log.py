#!/usr/bin/env python3
"""Synthetic `log.dat` for applications/unary_binary_imu_test in the format the reference's
program parses (/root/reference/applications/unary_binary_imu_test/main.cpp:243-280; the original
recording is elided from the reference tree):

    ODO <t> <v_right> <v_left>                wheel speeds of a differential drive (track 1.5 m)
    UTM <t> <easting> <northing> <altitude>   position fix
    IMU <t> <wx> <wy> <wz> <ax> <ay> <az>     body rates, specific force

A vehicle drives a smooth planar course for 40 s.  Conventions of the reference program: the body
frame has +y forward (its gyro dead-reckoning advances along y, main.cpp:83), z up, gravity vector
(0, 0, 9.8) with  v' = R a_m - g  (main.cpp:226, include/ba/Types.h:376-416), so a level IMU at rest
reads (0, 0, 9.8).  Deterministic (fixed seed).   python applications/unary_binary_imu_test/make_log.py
"""
import os

import numpy as np

rng = np.random.default_rng(20260)
T, IMU_HZ, ODO_HZ, UTM_HZ = 40.0, 50.0, 20.0, 1.0
G = 9.8
TRACK = 1.5
E0, N0, ALT0 = 478312.25, 4429765.5, 1612.0


def speed(t):
    return 4.0 + 1.5 * np.sin(0.21 * t)


def yaw_rate(t):
    return 0.12 * np.sin(0.17 * t) + 0.05


# integrate the planar course on a fine grid (body +y is forward: heading vector = R_z(yaw) e_y)
dt = 1e-3
ts = np.arange(0.0, T + 0.5, dt)
yaw = np.concatenate([[0.0], np.cumsum(yaw_rate(ts[:-1]) * dt)])
fwd = np.stack([-np.sin(yaw), np.cos(yaw)], 1)
pos = np.concatenate([[[0.0, 0.0]], np.cumsum(fwd[:-1] * speed(ts[:-1])[:, None] * dt, 0)])
vel = fwd * speed(ts)[:, None]
acc = np.gradient(vel, dt, axis=0)


def at(arr, t):
    return arr[int(round(t / dt))]


events = []
for k in range(int(T * IMU_HZ) + 1):
    t = k / IMU_HZ
    a_w = np.array([*at(acc, t), 0.0]) + np.array([0.0, 0.0, G])
    c, s = np.cos(at(yaw, t)), np.sin(at(yaw, t))
    a_b = np.array([c * a_w[0] + s * a_w[1], -s * a_w[0] + c * a_w[1], a_w[2]])   # R_z(yaw)^T a_w
    w_b = np.array([0.0, 0.0, yaw_rate(t)])
    a_b += rng.normal(0, 0.002, 3)
    w_b += rng.normal(0, 6e-5, 3)
    events.append((t, 1, "IMU %.6f %.9f %.9f %.9f %.9f %.9f %.9f" % (t, *w_b, *a_b)))
for k in range(int(T * ODO_HZ) + 1):
    t = k / ODO_HZ
    v, w = speed(t), yaw_rate(t)
    rr, rl = v + 0.5 * TRACK * w, v - 0.5 * TRACK * w
    events.append((t, 0, "ODO %.6f %.6f %.6f" % (t, rr + rng.normal(0, 0.01), rl + rng.normal(0, 0.01))))
for k in range(int(T * UTM_HZ) + 1):
    t = k / UTM_HZ + 0.005
    p = at(pos, t) + rng.normal(0, 0.3, 2)
    events.append((t, 2, "UTM %.6f %.4f %.4f %.4f" % (t, E0 + p[0], N0 + p[1], ALT0 + rng.normal(0, 0.5))))
events.sort(key=lambda e: (e[0], e[1]))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "log.dat")
with open(out, "w") as f:
    for _, _, line in events:
        f.write(line + "\n")
print("wrote", out, len(events), "lines")
