"""Constants of the OpenPose inference path (joint order, limb table, thresholds).

Mirrors the *inference* part of the reference's ``entity.py``:
  * ``JointType``  -- reference ``entity.py:9-45`` (18 COCO-style joints, same order/values)
  * ``params``     -- reference ``entity.py:71-105,126-151`` (inference keys of the pose, face and hand detectors;
                      training keys are out of the hot path; ``archs`` maps to native network names)

The same numbers are compiled into the HIP library (``csrc/pmx_common.h``, ``include/pose_mi355x.h``);
``tests/test_host.py`` checks both copies against each other and against the golden fixture generated from the
reference module (``tests/golden/host_fns.json``).
"""
from enum import IntEnum


class JointType(IntEnum):
    Nose = 0
    Neck = 1
    RightShoulder = 2
    RightElbow = 3
    RightHand = 4
    LeftShoulder = 5
    LeftElbow = 6
    LeftHand = 7
    RightWaist = 8
    RightKnee = 9
    RightFoot = 10
    LeftWaist = 11
    LeftKnee = 12
    LeftFoot = 13
    RightEye = 14
    LeftEye = 15
    RightEar = 16
    LeftEar = 17


_J = JointType

params = {
    # reference entity.py:59
    'downscale': 8,
    # reference entity.py:71-84
    'inference_img_size': 368,
    'inference_scales': [0.5, 1, 1.5, 2],
    'heatmap_size': 320,
    'gaussian_sigma': 2.5,
    'ksize': 17,
    'n_integ_points': 10,
    'n_integ_points_thresh': 8,
    'heatmap_peak_thresh': 0.05,
    'inner_product_thresh': 0.05,
    'limb_length_ratio': 1.0,
    'length_penalty_value': 1,
    'n_subset_limbs_thresh': 3,
    'subset_score_thresh': 0.2,
    # reference entity.py:85-105 -- 19 limbs, PAF channels (2i, 2i+1) = (x, y) of limb i
    'limbs_point': [
        [_J.Neck, _J.RightWaist],
        [_J.RightWaist, _J.RightKnee],
        [_J.RightKnee, _J.RightFoot],
        [_J.Neck, _J.LeftWaist],
        [_J.LeftWaist, _J.LeftKnee],
        [_J.LeftKnee, _J.LeftFoot],
        [_J.Neck, _J.RightShoulder],
        [_J.RightShoulder, _J.RightElbow],
        [_J.RightElbow, _J.RightHand],
        [_J.RightShoulder, _J.RightEar],
        [_J.Neck, _J.LeftShoulder],
        [_J.LeftShoulder, _J.LeftElbow],
        [_J.LeftElbow, _J.LeftHand],
        [_J.LeftShoulder, _J.LeftEar],
        [_J.Neck, _J.Nose],
        [_J.Nose, _J.RightEye],
        [_J.Nose, _J.LeftEye],
        [_J.RightEye, _J.RightEar],
        [_J.LeftEye, _J.LeftEar],
    ],
    # face / hand key-point detectors (reference entity.py:126-151)
    'face_inference_img_size': 368,
    'face_heatmap_peak_thresh': 0.1,
    'hand_inference_img_size': 368,
    'hand_heatmap_peak_thresh': 0.1,
    'fingers_indices': [
        [[0, 1], [1, 2], [2, 3], [3, 4]],
        [[0, 5], [5, 6], [6, 7], [7, 8]],
        [[0, 9], [9, 10], [10, 11], [11, 12]],
        [[0, 13], [13, 14], [14, 15], [15, 16]],
        [[0, 17], [17, 18], [18, 19], [19, 20]],
    ],
    # name -> network architecture handled by the native library (reference entity.py:50-54)
    'archs': {'posenet': 'posenet', 'facenet': 'facenet', 'handnet': 'handnet'},
}

N_JOINTS = len(JointType)          # 18
N_LIMBS = len(params['limbs_point'])  # 19
N_PAF_CH = 2 * N_LIMBS             # 38
N_HEAT_CH = N_JOINTS + 1           # 19 (last = background, dropped at pose_detector.py:78)
