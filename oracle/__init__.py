"""CPU oracle for the MobileNet-YOLO hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product
package (``mobilenet-yolo-pytorch_amd/``) never imports this package and fails
loudly when its HIP library is missing.

Contents
--------
``yolo_ref``   restatement of models/yolo_loss.py + utils/iou.py (decode, IoU,
               CIoU, target assignment, weighted MSE) in torch-CPU fp32.
``nms_ref``    ctypes wrapper over ``nms_ref.c`` — a C restatement of the
               torchvision CPU NMS kernel plus the per-class driver of
               utils/box.py:11-31.
``net_ref``    restatement of models/mobilenetv2.py + models/mbv2_yolo.py as a
               plain torch.nn CPU model with the reference's state_dict keys.
``procedural`` deterministic name-keyed weight fill shared by the golden
               generator and the parity tests.

Pinning: every function here is checked against fixtures captured from the
real reference (``tools/gen_golden.py`` -> ``tests/golden/*.npz``) except the
inner torchvision NMS kernel, which the reference does not vendor
(requirements.txt:3, unpinned) — that kernel is "parity unpinned" and is backed
by known-answer and property tests instead (tests/test_oracle_nms.py).
"""
