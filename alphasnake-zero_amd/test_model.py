"""One watched game of a saved model -- this build's counterpart of the reference's ``test_model.py`` script
(test_model.py:6-23): load ``models/<name>.h5``, print the net summary, empty ``replay.rep``, play one 11x11 4-snake game
(a single-game runner appends both boards of every tick to ``replay.rep``, mp_game_runner.py:26,52), then hand the file
to the reference's terminal viewer when its ``player.py`` is importable (the viewer is not part of this build).
Not a pytest file: ``python test_model.py [<model name>]``."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def play_one(model_name, height=11, width=11, snake_cnt=4):
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner
    net = AlphaNNet(model_name="models/" + model_name + ".h5")
    net.v_net.summary()
    open("replay.rep", "w").close()
    print("\nRunning games...")
    runner = MPGameRunner(height, width, snake_cnt)
    return runner.run(Agent(net)), runner


def main(argv):
    play_one(argv[0] if argv else input("\nEnter the model name:\n"))
    try:
        from player import Player
    except ImportError:
        print("replay.rep written (the reference's player.py shows it)")
        return
    input("\nHit Enter to watch the replay")
    Player().main()


if __name__ == "__main__":
    main(sys.argv[1:])
