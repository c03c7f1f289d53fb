"""Build profiles/<round>/<tag>_pmc_traffic.json (read by bench.py's roofline.traffic) from the fetch / write PMC
passes of tools/dbg/pmc_run.sh.   usage: pmc_traffic.py gpurun_out/pmc_<tag> out.json
FETCH_SIZE is in KiB and doubled (gfx950 counts a 128-B read request as 64 B, MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is in KiB and exact."""
import json, sys
d = sys.argv[1]
fetch, write = json.load(open(d + '/fetch.json')), json.load(open(d + '/write.json'))
names = {'gemm_p_kernel<1, false, CfgBig, true>': 'gemm_bf16_kernel<EPI_QKV_ROPE>',
         'gemm_p_kernel<2, false, CfgBig, false>': 'gemm_bf16_kernel<EPI_GATE_RES>',
         'gemm_p_kernel<3, false, CfgBig, false>': 'gemm_bf16_kernel<EPI_BIAS_GELU>',
         'gemm_p_kernel<9, false, CfgRow, false>': 'gemm_bf16_kernel<EPI_GATE_RES_LN>',
         'gemm_p_kernel<8, false, CfgRow, false>': 'bytenet: gemm_bf16_kernel<EPI_LN_GELU> (1x1, K = 1024)',
         'gemm_p_kernel<8, true, CfgRow, false>': 'bytenet: gemm_bf16_kernel<EPI_LN_GELU> (dilated conv, K = 2560)',
         'gemm_p_kernel<6, false, CfgBig, false>': 'bytenet: gemm_bf16_kernel<EPI_RES>',
         'ln_gelu_kernel<1024>': 'bytenet: ln_gelu_kernel<1024>',
         'attention_kernel': 'attention_kernel', 'attention_kernel<4>': 'attention_kernel', 'ln_mod_kernel<512>': 'ln_mod_kernel'}
out = {'_note': 'HBM bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs) over '
                'tools/dbg/dit_fwd.py at B=4096 (the bench shapes). FETCH_SIZE is in KiB and is doubled (gfx950 counts '
                '128-B read requests as 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE is in KiB and exact.'}
for rk, name in names.items():
    if rk not in fetch or rk not in write:
        continue
    f, w = fetch[rk], write[rk]
    fb, wb = f['FETCH_SIZE'] * 1024 * 2, w['WRITE_SIZE'] * 1024
    hit = f.get('TCC_HIT_sum', 0.0)
    miss, req = w.get('TCC_MISS_sum', 0.0), w.get('TCC_REQ_sum', 0.0)
    out[name] = dict(rocprof_name=rk, launches=f['n'], fetch_bytes=fb, write_bytes=wb, hbm_bytes=fb + wb,
                     l2_hit=(1.0 - miss / req) if req else None)
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, v in out.items():
    if k != '_note':
        print(k, {a: (round(b / 1e9, 3) if isinstance(b, float) and b > 1e6 else b) for a, b in v.items()})
