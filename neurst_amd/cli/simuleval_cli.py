"""simuleval_cli (neurst/cli/simuleval_cli.py) for the text agent: the reference forwards to SimulEval's own CLI; SimulEval is
not installed here, so this entry drives the registered agent with the local client loop of
neurst_amd/utils/simuleval_agents/simul_trans_text_agent.py over a source file (one sentence per line), writes the
hypotheses and reports the mean Average Lagging (+ the delays per sentence as JSON lines with --output).

    python -m neurst_amd.cli.simuleval_cli --agent simul_trans_text_agent --source src.txt --model-dir DIR --wait-k 3 \
           [--target ref.txt] [--output out_dir]
"""
import argparse
import json
import os

from neurst_amd.utils.simuleval_agents import AGENTS


def main(argv=None):
    from neurst_amd.utils.simuleval_agents import simul_trans_text_agent as A  # noqa: F401  (registers the agent)
    ap = argparse.ArgumentParser()
    ap.add_argument("--agent", default="simul_trans_text_agent")
    ap.add_argument("--data-type", default="text", choices=["text"], dest="data_type")
    ap.add_argument("--source", required=True)
    ap.add_argument("--target", default=None)
    ap.add_argument("--output", default=None)
    known, _ = ap.parse_known_args(argv)
    cls = AGENTS[known.agent]
    cls.add_args(ap)
    args = ap.parse_args(argv)
    agent = cls(args)
    results = []
    with open(args.source) as fp:
        for i, line in enumerate(fp):
            results.append(A.run_agent_on_sentence(agent, line.strip().split(), sentence_id=i))
    al = sum(r["average_lagging"] for r in results) / max(len(results), 1)
    if args.output:
        os.makedirs(args.output, exist_ok=True)
        with open(os.path.join(args.output, "instances.log"), "w") as fp:
            for i, r in enumerate(results):
                fp.write(json.dumps({"index": i, "prediction": " ".join(r["hypothesis"]), "delays": r["delays"],
                                     "average_lagging": r["average_lagging"]}) + "\n")
        with open(os.path.join(args.output, "hypothesis.txt"), "w") as fp:
            fp.write("\n".join(" ".join(r["hypothesis"]) for r in results) + "\n")
    print(json.dumps({"sentences": len(results), "AL": al}))
    return results


if __name__ == "__main__":
    main()
