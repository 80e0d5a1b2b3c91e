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


# joint order = reference entity.py:9-45 (value = index in this tuple)
JOINT_NAMES = ('Nose Neck RightShoulder RightElbow RightHand LeftShoulder LeftElbow LeftHand RightWaist RightKnee '
               'RightFoot LeftWaist LeftKnee LeftFoot RightEye LeftEye RightEar LeftEar').split()
JointType = IntEnum('JointType', [(name, i) for i, name in enumerate(JOINT_NAMES)])

# the 19 limbs as (from, to) joint names -- reference entity.py:85-105; PAF channels (2i, 2i+1) = (x, y) of limb i
_LIMBS = ('Neck-RightWaist RightWaist-RightKnee RightKnee-RightFoot Neck-LeftWaist LeftWaist-LeftKnee LeftKnee-LeftFoot '
          'Neck-RightShoulder RightShoulder-RightElbow RightElbow-RightHand RightShoulder-RightEar Neck-LeftShoulder '
          'LeftShoulder-LeftElbow LeftElbow-LeftHand LeftShoulder-LeftEar Neck-Nose Nose-RightEye Nose-LeftEye '
          'RightEye-RightEar LeftEye-LeftEar').split()

params = dict(
    downscale=8,                                  # reference entity.py:59
    # inference keys, reference entity.py:71-84
    inference_img_size=368, inference_scales=[0.5, 1, 1.5, 2], heatmap_size=320,
    gaussian_sigma=2.5, ksize=17,
    n_integ_points=10, n_integ_points_thresh=8,
    heatmap_peak_thresh=0.05, inner_product_thresh=0.05,
    limb_length_ratio=1.0, length_penalty_value=1,
    n_subset_limbs_thresh=3, subset_score_thresh=0.2,
    limbs_point=[[JointType[a], JointType[b]] for a, b in (pair.split('-') for pair in _LIMBS)],
    # COCO annotation order -> JointType (reference entity.py:106-124; the Neck has no COCO counterpart)
    coco_joint_indices=[JointType[n] for n in ('Nose LeftEye RightEye LeftEar RightEar LeftShoulder RightShoulder LeftElbow RightElbow '
                                               'LeftHand RightHand LeftWaist RightWaist LeftKnee RightKnee LeftFoot RightFoot').split()],
    # face / hand key-point detectors (reference entity.py:126-151)
    face_inference_img_size=368, face_heatmap_peak_thresh=0.1, face_crop_scale=1.5,
    # 68-point face polylines: jaw 0-16, brows 17-21 / 22-26, nose 27-30 / 31-35 (open), eyes 36-41 / 42-47 and lips 48-59 / 60-67 (closed)
    face_line_indices=([[i, i + 1] for a, b in ((0, 16), (17, 21), (22, 26), (27, 30), (31, 35)) for i in range(a, b)] +
                       [[i, i + 1 if i < b else a] for a, b in ((36, 41), (42, 47), (48, 59), (60, 67)) for i in range(a, b + 1)]),
    hand_inference_img_size=368, hand_heatmap_peak_thresh=0.1,
    fingers_indices=[[[0 if k == 0 else 4 * f + k, 4 * f + k + 1] for k in range(4)] for f in range(5)],
    # name -> network architecture handled by the native library (reference entity.py:50-54)
    archs={'posenet': 'posenet', 'facenet': 'facenet', 'handnet': 'handnet'},
)

N_JOINTS = len(JointType)          # 18
N_LIMBS = len(params['limbs_point'])  # 19
N_PAF_CH = 2 * N_LIMBS             # 38
N_HEAT_CH = N_JOINTS + 1           # 19 (last = background, dropped at pose_detector.py:78)
