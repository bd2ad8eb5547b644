"""Two saved models against each other -- this build's counterpart of the reference's ``test_pit.py`` script
(test_pit.py:7-65): 300 games each of model 1 alone against three snakes of model 2, the same with the roles swapped,
then a 2-snake duel; prints the win (and draw) rates in the reference's wording.  Not a pytest file: it asks for the two
model names (or takes them from the command line: ``python test_pit.py <model 1> <model 2>``)."""
import os
import sys
from time import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

HEIGHT = WIDTH = 11
HEALTH_DEC = 1
GAMES = 300


def tally(winner_ids, first_team_cnt):
    """-> (games the first team won, games the second team won, draws) (test_pit.py:27-31, 57-63)"""
    first = sum(1 for w in winner_ids if w is not None and w < first_team_cnt)
    draws = sum(1 for w in winner_ids if w is None)
    return first, len(winner_ids) - first - draws, draws


def match(first, second, snake_cnt, first_team_cnt, games=GAMES):
    from utils.pit_mp_game_runner import MPGameRunner
    print("\nRunning games...")
    t0 = time()
    ids = MPGameRunner(HEIGHT, WIDTH, snake_cnt, HEALTH_DEC, games).run(first, second, first_team_cnt)
    return tally(ids, first_team_cnt), time() - t0


def main(argv, games=GAMES):
    from utils.alpha_nnet import AlphaNNet
    from utils.pit_agent import Agent
    names = argv[:2] if len(argv) >= 2 else [input("\nEnter the model 1 name:\n"), input("\nEnter the model 2 name:\n")]
    agents = [Agent(AlphaNNet(model_name="models/" + n + ".h5")) for n in names]
    for me in (0, 1):                                                   # one against three, both ways round
        (won, _, draws), dt = match(agents[me], agents[1 - me], 4, 1, games)
        print("1v3 Win Rate of", names[me], won / games, "Draw Rate =", draws / games)
        print("Competing time", dt)
    (won, lost, _), dt = match(agents[0], agents[1], 2, 1, games)
    print("2v2 Win Rate of", names[0], won / games)
    print("2v2 Win Rate of", names[1], lost / games)
    print("Competing time", dt)


if __name__ == "__main__":
    main(sys.argv[1:])
