"""Small host-side math of the API mirror: pybullet's quaternion / view / projection conventions (numpy, float64).
Used by the facade for the reference's camera set-up (real_robots/envs/env.py:136-141,470-513) and for object poses given
as roll-pitch-yaw (real_robots/envs/robot.py:19-24).  The per-step arithmetic is in librealrobot_hip.so; the numpy
forward / inverse kinematics that CHECK the device IK live in oracle/kinematics.py (test infrastructure)."""
import numpy as np


def quat_from_euler(r, p, y):
    """pybullet.getQuaternionFromEuler (xyzw, rotation = Rz(y) Ry(p) Rx(r))."""
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy])


def look_at(eye, target, up):
    """pybullet.computeViewMatrix (right-handed look-at), row-major 4x4."""
    eye, target, up = (np.asarray(a, dtype=np.float64) for a in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.eye(4)
    V[0, :3], V[1, :3], V[2, :3] = s, u, -f
    V[:3, 3] = -V[:3, :3] @ eye
    return V


def view_from_yaw_pitch_roll(target, distance, yaw, pitch, roll):
    """pybullet.computeViewMatrixFromYawPitchRoll(..., upAxisIndex=2): eye = Rz(yaw) Ry(roll) Rx(pitch) (0,-d,0) + target."""
    y, p, r = np.radians([yaw, pitch, roll])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(r), 0, np.sin(r)], [0, 1, 0], [-np.sin(r), 0, np.cos(r)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    R = Rz @ Ry @ Rx
    eye = R @ np.array([0.0, -distance, 0.0]) + np.asarray(target, dtype=np.float64)
    return look_at(eye, target, R @ np.array([0.0, 0.0, 1.0]))


def perspective(fov_deg, aspect, near, far):
    """pybullet.computeProjectionMatrixFOV (OpenGL perspective), row-major 4x4."""
    ys = 1.0 / np.tan(np.radians(fov_deg) / 2)
    P = np.zeros((4, 4))
    P[0, 0], P[1, 1] = ys / aspect, ys
    P[2, 2], P[2, 3] = (near + far) / (near - far), 2 * near * far / (near - far)
    P[3, 2] = -1
    return P
