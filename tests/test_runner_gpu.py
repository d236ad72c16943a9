"""GPU (-m gpu): the reference's entry points end to end on the synthetic dataset —
main.py CLI -> Runner.train (2 steps) -> Runner.eval -> keypoint JSON -> OKS AP -> checkpoints."""
import json
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def test_main_train_then_eval(tmp_path, monkeypatch):
    from hupr_amd import main as hmain
    from hupr_amd.config_tree import CONFIG_DIR
    cfgd = yaml.safe_load(open(os.path.join(CONFIG_DIR, "mscsa_prgcn.yaml")))
    cfgd["DATASET"]["dataDir"] = "synthetic"
    cfgd["TRAINING"]["batchSize"] = 2
    cfgd["TRAINING"]["epochs"] = 1
    cfgd["TEST"]["batchSize"] = 2
    (tmp_path / "config").mkdir()
    yaml.safe_dump(cfgd, open(tmp_path / "config" / "tiny.yaml", "w"))
    (tmp_path / "logs").mkdir()
    (tmp_path / "visualization").mkdir()
    monkeypatch.chdir(tmp_path)
    hmain.main(["--config", "tiny.yaml", "--dir", "run0", "--synthetic_length", "4", "--max_steps", "2"])
    run = tmp_path / "logs" / "run0"
    ck = torch.load(run / "checkpoint.pth")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "accuracy"}
    assert len(ck["model_state_dict"]) == 255
    recs = json.load(open(run / "val_results.json"))
    assert len(recs) == 4 and len(recs[0]["keypoints"]) == 42 and recs[0]["score"] == 1.0
    assert all(0 <= v <= 252 for v in recs[0]["keypoints"][0::3])            # heat-map pixel * 4
    assert os.path.exists(run / "model_best.pth") and os.path.exists(run / "train_loss_list_0.json")
    # evaluation entry point reloads model_best and writes test_results.json
    hmain.main(["--config", "tiny.yaml", "--dir", "run0", "--synthetic_length", "4", "--eval"])
    assert len(json.load(open(run / "test_results.json"))) == 4
