"""cProfile of scripts/fit_e2e.py's fit_ call (development aid): python scripts/fit_e2e_profile.py M N K"""
import cProfile, pstats, sys, runpy
sys.argv = ["fit_e2e.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(str(__import__("pathlib").Path(__file__).resolve().parent / "fit_e2e.py"), run_name="__main__")
finally:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
