"""tools/first_contact_summary.py digests what tools/first_contact.sh records on a multi-GPU node; here it is fed transcripts recorded on
ONE GPU (profiles/r04: two and eight ranks of the 2-layer 7B-width model sharing a device) plus a made-up failing step."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import first_contact_summary as F      # noqa: E402


def test_digest_of_recorded_one_gpu_transcripts(tmp_path):
    two = open(os.path.join(ROOT, "profiles", "r04", "bench_gpus2_two_ranks_one_gpu.json")).read().strip().splitlines()[-1]
    eight = open(os.path.join(ROOT, "profiles", "r04", "bench_gpus8_eight_ranks_one_gpu.json")).read().strip().splitlines()[-1]
    # a full-size line with a committed prediction: the recorded two-rank line, relabelled as the full model with a prediction block
    j = json.loads(two)
    j["tp_predicted"] = {"exchanges_per_token": 65, "2": {"shard_step_ms": 0.30, "tok_s_zero_latency": 3333.3, "tok_s_2us_per_exchange": 2325.6, "tok_s_5us_per_exchange": 1600.0}}
    t = tmp_path / "t.txt"
    t.write_text("\n".join(["== pytest: two-GPU group", "2 skipped in 3.1s", "== bench --gpus 2 --config llama2_7b_L2", "rccl banner on stderr", two,
                            "== bench --gpus 8", eight, "== bench --gpus 2 (prediction)", json.dumps(j), "== bench that died", "Traceback (most recent call last):", "RuntimeError: boom"]) + "\n")
    text = "\n".join(F.main(str(t)))
    assert "== pytest: two-GPU group" in text and "no bench line; last output: 2 skipped in 3.1s" in text
    assert "tp.l2_tp_mode [3] (one hipGraph per token, one-shot peer-to-peer exchange)  devices [0, 0]  sharded True" in text
    assert "equal to the reference golden True" in text and "parity of the TIMED run: 64 of 64 steps checked, equal to the reference golden: True" in text
    # the eight-rank run on one GPU formed its group only at the third attempt: every failed step is listed
    assert text.count("note: ") >= 3 and "note: RCCL + peer-to-peer exchange: L2Error" in text and "the ranks met through files" in text
    assert "no committed prediction for this configuration" in text
    assert "against the prediction 0.3000 (a rank's shard alone)" in text and "us per exchange on top of it" in text
    assert "== bench that died" in text and "RuntimeError: boom" in text


def test_the_script_runs_its_steps_as_fresh_processes_in_the_stated_order():
    sh = open(os.path.join(ROOT, "tools", "first_contact.sh")).read()
    order = [sh.index(k) for k in ('-k "two_gpu or other_than_the_threads"', "--gpus 2 --config llama2_7b_L2", "for G in 2 4 8", "L2_TP_FENCED=1 step", "first_contact_summary.py")]
    assert order == sorted(order)
    assert "exec " not in sh and "HSA_ENABLE_IPC_MODE_LEGACY=0" in sh
