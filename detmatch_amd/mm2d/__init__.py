"""2D branch of DetMatch: Faster R-CNN (ResNet-50 caffe + FPN + RPN + RoIAlign + Shared2FC).

The reference takes all of this from mmdet 2.14.0 / mmcv-full 1.3.16 (un-vendored third-party
packages, absent from /root/reference): the modules here restate their published behaviour for
the configuration at configs/detmatch/001/detmatch/split_0.py:39-99,440-478,507-529 —
PARITY UNPINNED.  Parameter names follow mmdet so checkpoints map 1:1.
"""
from .faster_rcnn import FasterRCNN  # noqa: F401
