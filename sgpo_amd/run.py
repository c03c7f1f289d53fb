"""Run one of the reference's drivers, unchanged, on the MI355X drop-in:

    cd /path/to/SGPO && python -m sgpo_amd.run pareto.py data=CreiLOV model=mdlm algorithm=cls_guidance_discrete ...

`python pareto.py` by itself puts the script's directory at sys.path[0], ahead of PYTHONPATH, so the reference's own
`models/`, `sampling/` ... would win there.  This launcher fixes the order -- drop-in first, then the driver's
directory -- and runs the script as `__main__` (hydra finds `configs/` next to the script as usual).  Modules the
drop-in does not mirror fall through to the reference (sgpo_amd/dropin/_fallthrough.py).
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print(f'sgpo_amd.run: no such driver: {argv[0]}', file=sys.stderr)
        return 2
    import sgpo_amd
    dropin = sgpo_amd.install_dropin()
    root = os.path.dirname(script)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (dropin, root)]
    sys.path[0:0] = [dropin, root]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
