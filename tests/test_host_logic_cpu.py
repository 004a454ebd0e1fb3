"""CPU: host-side logic of the stages around the path (no kernels run here) — input-prep packing, mAP list packing and
bookkeeping, the reference-shaped return contracts, and the loud failure without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_batchprep_pack_layout_and_validation():
    from mobilenet_yolo_pytorch_amd import prep, synthetic
    bp = prep.BatchPrep([[352, 352], [320, 320]], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], device="cpu")
    imgs = synthetic.photos([(5, 7), (4, 4), (9, 3)], seed=1)
    stage, desc, mh, mw = bp.pack(imgs)
    assert (mh, mw) == (9, 7) and desc.dtype.itemsize == 16                       # mny_image_desc: int64 + 2 x int32
    assert all(int(o) % 16 == 0 for o in desc["offset"]) and list(desc["h"]) == [5, 4, 9] and list(desc["w"]) == [7, 4, 3]
    buf = stage.numpy()
    for d, im in zip(desc, imgs):
        assert np.array_equal(buf[d["offset"]:d["offset"] + im.size].reshape(im.shape), im)
    assert tuple(bp.choose_size()) in ((352, 352), (320, 320))
    with pytest.raises(ValueError):
        bp.pack([np.zeros((4, 4, 3), np.float32)])
    with pytest.raises(ValueError):
        bp([])


def test_map_list_packing_and_cpu_rejection():
    from mobilenet_yolo_pytorch_amd import evalmap
    ts = [torch.zeros(2, 4), torch.zeros(0, 4), torch.ones(3, 4)]
    flat, off = evalmap._pack_list(ts, 4, torch.device("cpu"))
    assert flat.shape == (5, 4) and off.tolist() == [0, 2, 2, 5] and off.dtype == torch.int32
    flat, off = evalmap._pack_list([torch.zeros(0), torch.ones(2)], 0, torch.device("cpu"))
    assert flat.shape == (2,) and off.tolist() == [0, 0, 2]
    z = torch.zeros(0)
    with pytest.raises(RuntimeError, match="CUDA"):
        evalmap.map_eval(torch.zeros(0, 4), z, z, torch.zeros(1, dtype=torch.int32), torch.zeros(0, 4), z, z, torch.zeros(1, dtype=torch.int32), 3)
    assert evalmap.adjust_confidence(10, 31, 0.1) == pytest.approx(0.11) and evalmap.adjust_confidence(10, 5, 0.01) == 0.01


def test_evaluator_bookkeeping():
    from mobilenet_yolo_pytorch_amd import evalmap
    ev = evalmap.Evaluator(["background", "a", "b"])
    ev.add([torch.zeros(3, 7), None], [torch.zeros(2, 5), np.zeros((0, 5), np.float32)])
    ev.add_packed(torch.zeros(4, 7), [1, 3], [torch.zeros(1, 5), torch.zeros(2, 5)])
    assert ev.row_counts == [3, 0, 1, 3] and ev.tg_counts == [2, 0, 1, 2] and ev.gt_box == 5 and ev.pred_box == 7


def test_seg_config_contract_without_gpu():
    from mobilenet_yolo_pytorch_amd import yolo
    man = json.load(open(os.path.join(G, "state_keys_bdd100k.json")))
    m = yolo(man["config"])
    assert m.has_seg and m.seg_num_classes == 2 and m.graph.seg_out is not None
    assert not yolo(json.load(open(os.path.join(G, "state_keys_voc.json")))["config"]).has_seg
    with pytest.raises(Exception):                 # CPU tensors: the HIP path is the only path
        m.train()(torch.zeros(1, 3, 96, 96), [torch.zeros(0, 5)], torch.zeros(1, 6, 6, 2))
