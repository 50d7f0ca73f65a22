"""Debug aid for tests/test_gpu_team_ranks.py: run the rank-process scenarios for the given modes and say which modes fail.
   python tools/dbg_modes.py WORLD N DECOMP [mode,mode,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world, n, decomp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    modes = []
    for m in (sys.argv[4].split(',') if len(sys.argv) > 4 else []):          # "mode*5" = five fresh teams of that mode
        base, _, rep = m.partition('*')
        modes += [base] + [f"{base}#{i}" for i in range(1, int(rep or 1))]
    os.environ["MI355CG_TEST_EXTRA_MODES"] = ",".join(m for m in modes if '#' in m)
    import test_gpu_team_ranks as T
    modes = modes or list(T.MODES)
    ref = T._reference(n)
    parts = T._run(world, n, decomp, modes)
    for mode in dict.fromkeys(modes):
        try:
            T._check(parts, ref, world, [mode])
            print(mode, "OK")
        except AssertionError as e:
            p = parts[0]
            print(mode, "FAIL", "rel", p[f"{mode}/rel"][:3], "ref", ref["rel"].iterations, "msg", p[f"{mode}/msg"][:3], "ref", ref["msg"].iterations,
                  "stop0", p[f"{mode}/stop0"], "stop1", p[f"{mode}/stop1"], "fixed", p[f"{mode}/fixed"], ref["fixed"].r_norm2, "desc", [list(q[f"{mode}/desc"][[0, 2, 4, 8]]) for q in parts], str(e)[:60].replace("\n", " "))


if __name__ == "__main__":
    main()
