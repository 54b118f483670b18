# yolov6s6: inference-path model description (the fields build_network reads) of the P6 assembly
# EfficientRep6 + RepBiFPANNeck6 + four-level LP head.  Values follow the reference's configs/yolov6s6.py;
# solver / augmentation sections belong to training and are out of scope here.
_BACKBONE_C = [64, 128, 256, 512, 768, 1024]
_NECK_C = [512, 256, 128, 256, 512, 1024]

model = dict(
    type='YOLOv6s6', pretrained=None, depth_multiple=0.33, width_multiple=0.50,
    backbone=dict(type='EfficientRep6', num_repeats=[1, 6, 12, 18, 6, 6], out_channels=_BACKBONE_C, fuse_P2=True, cspsppf=True),
    neck=dict(type='RepBiFPANNeck6', num_repeats=[12] * 6, out_channels=_NECK_C),
    head=dict(type='EffiDeHead', in_channels=[128, 256, 512, 1024], num_layers=4, anchors=1, strides=[8, 16, 32, 64],
              use_dfl=False, reg_max=0, iou_type='giou'),
)
training_mode = 'repvgg'   # tools/train.py:84-85 default: RepVGGBlock + ReLU backbone/neck
