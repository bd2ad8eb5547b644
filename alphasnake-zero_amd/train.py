"""Start (or resume) the endless generation loop: self-play on the MI355X engine, ``log.csv``, fit, ``models/<name><n>.h5``
-- this build's counterpart of the reference's ``train.py`` entry script (train.py:6-40).  The reference's own file
also runs unchanged on this package (its ``import tensorflow`` finds the TPU-less stand-in next to ``utils/``,
INTEGRATION.md section 1); this one drops the TPU probe and adds the several-GPU start.

Same settings as train.py:6-13 (11x11, 4 snakes, 256 games, depth 8, breadth 128, lr 1e-4, decay 0.98), the same two
questions (:29-30; they may be answered on the command line: ``python train.py <name> <starting generation>``), the same
start: generation 0 creates a fresh net and saves ``<name>0`` (:31-33), a later start loads ``models/<name><n>.h5`` and
applies the decay that many times (:34-36).

Several GPUs: ``python -m torch.distributed.run --nproc-per-node N train.py <name> <generation>`` -- one process per GPU;
the games are cut into per-rank shards and the fit runs data-parallel over RCCL
(``utils.alpha_snake_zero_trainer``); rank 0 writes the files.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

SETTINGS = dict(game_board_height=11, game_board_width=11, number_of_snakes=4, self_play_games=256, max_MCTS_depth=8,
                max_MCTS_breadth=128, initial_learning_rate=0.0001, learning_rate_decay=0.98)


def join_ranks():
    """one process per GPU when started by torchrun; -> (rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    dist.init_process_group("nccl")                                    # RCCL
    return dist.get_rank(), world


def start(name, generation, max_iterations=None, **overrides):
    from utils.alpha_nnet import AlphaNNet
    from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
    s = dict(SETTINGS, **overrides)
    rank, world = join_ranks()
    h, w = s["game_board_height"], s["game_board_width"]
    lr = s["initial_learning_rate"]
    if rank == 0:
        os.makedirs("models", exist_ok=True)
    if generation == 0:
        net = AlphaNNet(input_shape=(2 * h - 1, 2 * w - 1, 3))
        if world > 1:                                                  # every rank starts from rank 0's draw
            import torch
            import torch.distributed as dist
            ws = [torch.as_tensor(a).cuda() for a in net.v_net.get_weights()]
            for t in ws:
                dist.broadcast(t, 0)
            net.v_net.set_weights([t.cpu().numpy() for t in ws])
        if rank == 0:
            net.save(name + "0")
    else:
        net = AlphaNNet(model_name="models/" + name + str(generation) + ".h5")
        lr *= s["learning_rate_decay"] ** generation
    trainer = AlphaSnakeZeroTrainer(s["self_play_games"], s["max_MCTS_depth"], s["max_MCTS_breadth"], lr,
                                    s["learning_rate_decay"], h, w, s["number_of_snakes"], None)
    return trainer.train(net, name=name, iteration=generation, max_iterations=max_iterations)


def main(argv):
    if len(argv) >= 2:
        name, generation = argv[0], int(argv[1])
    else:
        name = input("Enter the model name (not including the generation number nor \".h5\"):\n")
        generation = int(input("Enter the starting generation (0 for creating a new model):\n"))
    start(name, generation)


if __name__ == "__main__":
    main(sys.argv[1:])
