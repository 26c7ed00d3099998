"""Digest of tools/first_contact.sh: one block per step of the script from the JSON lines bench.py printed (and the pytest tail), in
the order a reader needs them on the first multi-GPU node -- how the group was formed (l2_tp_mode of every rank, the note of every
formation step that failed), whether the timed tokens are the reference's, and the measured step against the single-GPU prediction
(`tp_predicted`: a rank's shard alone on a GPU + 0 / 2 / 5 us per exchange).

  python tools/first_contact_summary.py <transcript>      # lines "== <label>" followed by what that step printed
"""
import json
import sys

MODES = {0: "not tensor parallel", 1: "eager launches + RCCL collectives", 2: "one hipGraph per token, RCCL collectives captured",
         3: "one hipGraph per token, one-shot peer-to-peer exchange", 4: "loopback test group", 5: "shard-timing context"}


def digest(label, lines):
    out = ["== " + label]
    js = [ln for ln in lines if ln.startswith("{") and '"metric"' in ln]
    if not js:
        tail = [ln for ln in lines if ln.strip()][-3:]
        out.append("   no bench line; last output: " + " | ".join(tail) if tail else "   no output")
        return out
    j = json.loads(js[-1])
    tp = j.get("tp") or {}
    out.append("   %s  n_gpus %s  parallelism %s  value %.2f %s  ms_per_step %.4f" % (
        (j.get("config") or {}).get("workload", "?").split(" batch-1")[0], j.get("n_gpus"), (j.get("config") or {}).get("parallelism"), j.get("value") or 0.0, j.get("unit"), j.get("ms_per_step") or 0.0))
    modes = tp.get("l2_tp_mode")
    out.append("   tp.l2_tp_mode %s (%s)  devices %s  sharded %s" % (modes, "; ".join(MODES.get(m, "?") for m in (modes or [])), tp.get("devices"), tp.get("sharded")))
    for f in tp.get("formation") or []:      # round 6: every way of forming the group is a stage of fresh worker processes with a deadline per phase
        out.append("   stage %-8s %s in %5.1f s%s%s" % (f.get("stage"), "ok    " if f.get("ok") else "FAILED", f.get("seconds") or 0.0,
                                                       ("  timed out on ranks %s" % f["timed_out_ranks"]) if f.get("timed_out_ranks") else "", ("  -- " + f["why"]) if f.get("why") else ""))
    if j.get("failed"):
        out.append("   NO MEASUREMENT: %s" % j.get("error"))
    proof = tp.get("proved_before_timing")
    if proof:
        out.append("   proved before timing: tokens %s  same on every rank %s  equal to the reference golden %s" % (proof.get("tokens"), proof.get("same_on_every_rank"), proof.get("equals_reference_golden")))
    if j.get("note"):
        for step in j["note"].split("; "):
            out.append("   note: " + step)
    par = j.get("parity") or {}
    out.append("   parity of the TIMED run: %s of %s steps checked, equal to the reference golden: %s%s" % (
        par.get("steps_checked"), par.get("steps_timed"), par.get("equal_to_reference_golden"), ("  (first mismatch at step %s)" % par["first_mismatch"]) if par.get("first_mismatch") is not None else ""))
    pred = (j.get("tp_predicted") or {})
    row = pred.get(str(j.get("n_gpus")))
    if row and "shard_step_ms" in row:
        ms = j["ms_per_step"]
        per_x = (ms - row["shard_step_ms"]) * 1e3 / max(pred.get("exchanges_per_token", 65), 1) if "exchanges_per_token" in pred else (ms - row["shard_step_ms"]) * 1e3 / 65.0
        out.append("   measured %.4f ms per token against the prediction %.4f (a rank's shard alone); %.2f us per exchange on top of it; predicted tok/s at 0 / 2 / 5 us per exchange: %s / %s / %s" % (
            ms, row["shard_step_ms"], per_x, row.get("tok_s_zero_latency"), row.get("tok_s_2us_per_exchange"), row.get("tok_s_5us_per_exchange")))
    elif j.get("n_gpus", 1) > 1:
        out.append("   no committed prediction for this configuration (profiles/tp_predicted.json holds the full Llama-2-7B shape only)")
    return out


def main(path):
    label, buf, blocks = None, [], []
    for ln in open(path, errors="replace").read().splitlines():
        if ln.startswith("== "):
            if label is not None:
                blocks.append((label, buf))
            label, buf = ln[3:].strip(), []
        elif label is not None:
            buf.append(ln)
    if label is not None:
        blocks.append((label, buf))
    text = []
    for lb, b in blocks:
        text += digest(lb, b)
    print("\n".join(text))
    return text


if __name__ == "__main__":
    main(sys.argv[1])
