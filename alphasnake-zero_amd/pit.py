"""Champion ladder: every new generation of a model plays the reigning champion and takes the title when it scores above
the threshold -- this build's counterpart of the reference's ``pit.py`` entry script (pit.py:7-62) and the writer of
``pit.txt``.  The reference's own, unmodified ``pit.py`` also runs on this package's
``utils``; this file is for users who do not carry the reference along.

Behaviour kept:
  * 1 000 games of 2 snakes on 11x11, health decrement 1, one snake per side (pit.py:7-11, 21, 33-35);
  * the challenger's score: a drawn game (winner ``None``) is half a point for each side, a winner id below the
    champion's snake count is the champion's game (pit.py:37-45); the title changes at ``score > 0.51`` (:46);
  * the three ``pit.txt`` line formats, byte for byte (:17, :49-50, :54-55, the reference's spelling included);
  * a missing ``models/<name><n>.h5`` is not an error: the ladder polls for it every 10 s (:56-62, ``OSError`` from the
    loader is the signal).
Additions, inert by default: ``max_challengers`` / ``max_polls`` end the otherwise endless loop (tests, batch jobs), and
the prompts can be answered on the command line: ``python pit.py <name> <first champion generation>``.
"""
import os
import sys
from time import sleep, time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

PIT_GAMES = 1000
THRESHOLD = 0.51
HEIGHT = WIDTH = 11
SNAKE_CNT = 2


def challenger_score(winner_ids, champion_snake_cnt):
    """share of the points the challenger (snake ids >= champion_snake_cnt) took (pit.py:37-45)"""
    won = lost = 0.0
    for w in winner_ids:
        if w is None:
            won += 0.5
            lost += 0.5
        elif w < champion_snake_cnt:
            lost += 1.0
        else:
            won += 1.0
    return won / (won + lost)


def verdict_line(tag, score, threshold=THRESHOLD):
    """the pit.txt line for challenger ``tag`` (pit.py:49-50, 54-55)"""
    if score > threshold:
        return tag + " beats the previouse champion. score = " + str(score) + ". It is the new champion!\n"
    return tag + " failed to beat the previouse champion. score = " + str(score) + ".\n"


def _note(path, line):
    with open(path, "a") as f:
        f.write(line)


def ladder(model_name, generation, pit_games=PIT_GAMES, threshold=THRESHOLD, height=HEIGHT, width=WIDTH,
           snake_cnt=SNAKE_CNT, log_path="pit.txt", poll_seconds=10, max_challengers=None, max_polls=None):
    """-> [(generation, score, took_the_title)] of the challengers that played"""
    from utils.alpha_nnet import AlphaNNet
    from utils.pit_agent import Agent
    from utils.pit_mp_game_runner import MPGameRunner

    def load(g):
        return AlphaNNet(model_name="models/" + model_name + str(g) + ".h5")

    champion = Agent(load(generation))
    _note(log_path, model_name + str(generation) + " is set to be the baseline champion.\n")
    champion_snake_cnt = snake_cnt // 2
    generation += 1
    played, polls, announce_wait = [], 0, False
    while max_challengers is None or len(played) < max_challengers:
        try:
            net = load(generation)
        except OSError:                                  # not written yet: wait for the trainer (pit.py:56-62)
            if announce_wait:
                print("Waiting for", model_name + str(generation) + "...")
                announce_wait = False
            polls += 1
            if max_polls is not None and polls >= max_polls:
                break
            sleep(poll_seconds)
            continue
        announce_wait = True
        tag = model_name + str(generation)
        print("A new challenger,", tag)
        challenger = Agent(net)
        t0 = time()
        runner = MPGameRunner(height, width, snake_cnt, 1, pit_games)
        print("Running games...")
        score = challenger_score(runner.run(champion, challenger, champion_snake_cnt), champion_snake_cnt)
        took = score > threshold
        if took:
            champion = challenger
        _note(log_path, verdict_line(tag, score, threshold))
        print("Competing time", time() - t0)
        played.append((generation, score, took))
        generation += 1
    return played


def main(argv):
    if len(argv) >= 2:
        model_name, generation = argv[0], int(argv[1])
    else:
        model_name = input("Enter the model name (not including the generation number nor \".h5\"):\n")
        generation = int(input("Enter the starting generation (the first champion):\n"))
    ladder(model_name, generation)


if __name__ == "__main__":
    main(sys.argv[1:])
