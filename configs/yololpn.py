# yololpn: inference-path model description (the fields build_network reads).
# Values follow the reference's configs/yololpn.py; solver / augmentation
# sections belong to training and are out of scope here.
_BACKBONE_C = [64, 128, 256, 512, 1024]
_NECK_C = [256, 128, 128, 256, 256, 512]

model = dict(
    type='YOLOv6n', pretrained=None, depth_multiple=0.33, width_multiple=0.25,
    backbone=dict(type='EfficientRep', num_repeats=[1, 6, 12, 18, 6], out_channels=_BACKBONE_C, fuse_P2=True, cspsppf=True),
    neck=dict(type='RepBiFPANNeck', num_repeats=[12, 12, 12, 12], out_channels=_NECK_C),
    head=dict(type='EffiDeHead', in_channels=[128, 256, 512], num_layers=3, strides=[8, 16, 32],
              use_dfl=False, reg_max=0, iou_type='siou'),
)
training_mode = 'repvgg'   # tools/train.py:84-85 default: RepVGGBlock + ReLU backbone/neck
