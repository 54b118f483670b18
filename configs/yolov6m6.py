# yolov6m6: inference-path model description (the fields build_network reads) of the P6 assembly
# CSPBepBackbone_P6 + CSPRepBiFPANNeck_P6 + four-level LP head with DFL.  Values follow the reference's
# configs/yolov6m6.py; solver / augmentation sections belong to training and are out of scope here.
_BACKBONE_C = [64, 128, 256, 512, 768, 1024]
_NECK_C = [512, 256, 128, 256, 512, 1024]

model = dict(
    type='YOLOv6m6', pretrained=None, depth_multiple=0.60, width_multiple=0.75,
    backbone=dict(type='CSPBepBackbone_P6', num_repeats=[1, 6, 12, 18, 6, 6], out_channels=_BACKBONE_C, csp_e=float(2) / 3,
                  fuse_P2=True),
    neck=dict(type='CSPRepBiFPANNeck_P6', num_repeats=[12] * 6, out_channels=_NECK_C, csp_e=float(2) / 3),
    head=dict(type='EffiDeHead', in_channels=[128, 256, 512, 1024], num_layers=4, anchors=1, strides=[8, 16, 32, 64],
              use_dfl=True, reg_max=16, iou_type='giou'),
)
training_mode = 'repvgg'   # tools/train.py:84-85 default: RepVGGBlock + ReLU backbone/neck
