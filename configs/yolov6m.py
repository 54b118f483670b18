# yolov6m: inference-path model description (the fields build_network reads).
# Values follow the reference's configs/yolov6m.py; solver / augmentation
# sections belong to training and are out of scope here.
_BACKBONE_C = [64, 128, 256, 512, 1024]
_NECK_C = [256, 128, 128, 256, 256, 512]

model = dict(
    type='YOLOv6m', pretrained=None, depth_multiple=0.6, width_multiple=0.75,
    backbone=dict(type='CSPBepBackbone', num_repeats=[1, 6, 12, 18, 6], out_channels=_BACKBONE_C, fuse_P2=True, csp_e=float(2) / 3),
    neck=dict(type='CSPRepBiFPANNeck', num_repeats=[12, 12, 12, 12], out_channels=_NECK_C, csp_e=float(2) / 3),
    head=dict(type='EffiDeHead', in_channels=[128, 256, 512], num_layers=3, strides=[8, 16, 32],
              use_dfl=True, reg_max=16, iou_type='giou'),
)
training_mode = 'repvgg'   # tools/train.py:84-85 default: RepVGGBlock + ReLU backbone/neck
