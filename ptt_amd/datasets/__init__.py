"""Data-side mirror, limited to what the sequential tracking loop's pre/post-processing needs:
ptt.datasets.kitti.kitti_tracking_utils. Dataset loading, augmentation and IO stay out of scope."""
