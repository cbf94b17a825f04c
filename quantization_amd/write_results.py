"""Perplexity result writer with the reference's interface and on-disk layout
(/root/reference/quant_utils/write_results.py:3-27).  Pure host code: nothing to accelerate."""
import os


def write_results(ppl: float, model_name: str, dataset: str, wq_bits: int, wq_datatype: str, wq_groupsize: int):
    """Writes '<dataset> perplexity: <ppl> ' to
    ./results_quant/<dataset>/<org>__<model>/[w_<bits>_gs_<gs|none>/]<datatype>.txt
    (the w_..._gs_... level is skipped for the unquantized fp32 / fp16 runs)."""
    gs = "none" if wq_groupsize <= 0 else wq_groupsize
    org, _, rest = model_name.partition("/")
    model_dir = f"{org}__{rest.split('/')[0]}" if rest else f"{org}__"
    parts = ["./results_quant", f"{dataset}", model_dir]
    if wq_datatype not in ("fp32", "fp16"):
        parts.append(f"w_{wq_bits}_gs_{gs}")
    out_dir = "/".join(parts)
    os.makedirs(out_dir, exist_ok=True)
    with open(f"{out_dir}/{wq_datatype}.txt", "w") as f:
        f.write(f"{dataset} perplexity: {ppl} \n")
    print("Successfully written results. \n\n")
