"""MI355X-native OpenPose inference path (drop-in for the reference's PoseDetector hot path)."""
