"""MobileNetV3-Large-YOLO drop-in (reference: models/mobilenetv3.py:77-136 + models/mbv3_yolo.py:97-145).

Same contract as `model.yolo`; the reference file cannot even be imported as shipped (it imports the
non-existent `models.voc.*`, mbv3_yolo.py:5-6) and needs a local weight file in its constructor (:104) —
this class builds from random init / `load_state_dict` instead (Q11).  Quirks kept: the "SE" module gates
per pixel (no pooling), `connect_for_S16` is applied twice with shared weights (Q12), PartAdd concatenates
the upper 160 channels of the upsampled S32."""
from .model import yolo as _yolo_base


class yolo(_yolo_base):
    ARCH = "mbv3"
