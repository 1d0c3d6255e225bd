"""Data-parallel launch of the UNCHANGED reference driver: one process per GPU, gradients all-reduced over RCCL / xGMI.

    PYTHONPATH=<repo>/integration/climb_shim:<repo> python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \\
        --master-addr 127.0.0.1 --master-port 29500 <repo>/integration/climb_torchrun.py <CLiMB>/src/train/train_upstream_continual_learning.py \\
        --encoder_name vilt --pretrained_model_name dandelin/vilt-b32-mlm --ordered_cl_tasks vqa,nlvr2,snli-ve,vcr --cl_algorithm ewc \\
        --ewc_fisher_sample_percentage 0.01 --ewc_loss_weight 100 --climb_data_dir /data/datasets/MCL/ --do_train --do_eval \\
        --output_dir /data/experiments/MCL/ --batch_size 64

What this adds around the driver (REF/train/train_upstream_continual_learning.py has no distributed code):
  * binds the process to GPU LOCAL_RANK (the driver's `torch.device("cuda")`, :39-40, then means that GPU) and joins the RCCL job;
  * checkpoints AND results files are written by rank 0 only, the latter through a temporary file + os.replace (`climb_amd.parallel.single_writer_io`,
    a context manager held around the driver call); ranks > 0 log warnings only.
Everything else happens inside the package the driver already calls: the trainers build rank-sharded loaders (`--batch_size` stays the
GLOBAL batch, so lr / warm-up / steps per epoch are the single-GPU run's), attach the gradient all-reducer in `train()`, all-reduce the
validation score in `eval()`; EWC's Fisher pass runs replicated and is broadcast from rank 0; the replay memory draws the same indices on
every rank and each rank replays its share.  The seed (`--seed`, REF/utils/seed_utils.py) must be the same on every rank -- it is, the
command line is.  BASELINE.json configs[3] / configs[4] are this command with `--cl_algorithm ewc` / `experience_replay`."""
import logging
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: climb_torchrun.py <driver.py> [driver arguments ...]")
    from climb_amd import parallel
    rank, world, device = parallel.init_data_parallel(os.environ.get("CLIMB_AMD_DP_BACKEND"))
    if rank != 0:
        logging.getLogger().setLevel(logging.WARNING)
    driver = sys.argv[1]
    sys.argv = [driver] + sys.argv[2:]
    # NOT from <CLiMB>/src as the working directory: the driver puts '.' first on sys.path (:19) and the reference's own `modeling` /
    # `cl_algorithms` packages there would shadow the shim
    try:
        with parallel.single_writer_io():          # torch.save / os.makedirs / json.dump-to-file are N-process-safe for the driver call, and only for it
            runpy.run_path(driver, run_name="__main__")
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
