# how much of a sweep is the run-start row loads (timing-only builds: garbage results)
VARIANTS="BASE KRUNNOLD KRUNNOLD_KRUNNOST BASE" tools/sweep_variants.sh run
