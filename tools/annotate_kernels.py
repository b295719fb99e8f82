"""Per-kernel roofline table for profiles/: the lines of tools/bench_kernels.py + one line of explanation for every row under half of the
HBM peak + the rows measured inside the generic iteration (no C-ABI builder for row-pattern tables).
usage: annotate_kernels.py raw.jsonl previous_table.jsonl > profiles/rNN_kernels.jsonl"""
import json
import sys

NOTES = {
    "grad2d_fwd (N/2)^2 x 4 label_first": "round 6: 16 bytes per lane on the label-first layout too (grad_fwd_lf_vec_kernel); the scalar kernel measured 0.441",
    "grad2d_adj (N/2)^2 x 4 label_first": "round 6: grad_adj_lf_vec_kernel; the scalar kernel measured 0.412",
    "diags_fwd 29 diagonals": "compulsory bytes count x once, the kernel reads it 29 times per row (L1 / L2 hits, LDS-staged band table): bound by the 29 load + "
                              "fma issue slots per row and the 64 bytes per clock a CU's L1 delivers, not by HBM (5 diagonals: 0.57)",
    "csr_spmv_acc 30 nnz/row": "16 lanes cooperate on a row (30 value + 30 index reads, a gather of 30 operand values, a 16-lane reduction): the gathered "
                               "operand lines are re-read across rows from L2",
    "sparse_kron_id (12 x 16, d = n/16)": "round 6: kron_id_vec_kernel (16 bytes per lane, the rows of S walked by the lane that owns the offsets; S through "
                                          "scalar loads): 0.18 -> 0.53; what is left: 36 operand loads per 12 stores, two thirds of them cache hits",
    "id_kron_sparse (12 x 16, d = n/16)": "round 6: id_kron_lds_kernel (operand tiles of 4096 elements and S staged in LDS, 16-byte loads and stores): 0.15 -> "
                                          "0.34; a tile is loaded, then computed, with a barrier between -- the loads of the next tile are not in flight "
                                          "during the products",
    "prox_epi_quad dim 3": "arithmetic-bound, not a stream: the projection solves a cubic per element (pow, acos, cos in device libm, helper.hpp:44-105): "
                           "~600 VALU instructions per element",
}


def main(raw, previous):
    for line in open(raw):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        if d["kernel"] in NOTES:
            d["note"] = NOTES[d["kernel"]]
        elif d.get("frac_of_8TBps") is not None and d["frac_of_8TBps"] < 0.5:
            d["note"] = "UNEXPLAINED: under half of the HBM peak"
        print(json.dumps(d))
    for line in open(previous):        # rows that come from counters of the generic iteration, not from bench_kernels.py
        d = json.loads(line)
        if d.get("source"):
            print(json.dumps(d))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
