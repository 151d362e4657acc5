"""Named prompt styles of the offline evaluation (``--prompt-type``).

The reference keeps one literal (question template, answer template, separator) triple per style in
``evaluation/utils.py:78-245`` and assembles prompts in ``construct_prompt`` (``:250-330``). Here a style is put
together from parts - a chat wrapper (ChatML, DeepSeek-R1 turns, Mistral, Alpaca-like headers, plain), an optional
system line, an instruction that goes before or after the question, and what the assistant turn starts with - so the
styles that differ only in one of these share the rest. The names are the reference's, so evaluation scripts and
result tables carry over.

    style("qwen25-math-cot").render("What is 1+1?")
    style("cot").render("What is 1+1?", shots=[("2+2?", "4"), ("3+3?", "6")])
    build_prompt(row, data_name="gsm8k", prompt_type="deepseek-math", num_shots=0)
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

BOXED = "Please reason step by step, and put your final answer within \\boxed{}."
BOXED_CHOICE = ("Please reason step-by-step and put your choice letter without any other text with \\boxed{} in the "
                "end.")
ANSWER_IS = 'put your final answer at the end with "The answer is: ".'
R1_ZERO_PREAMBLE = (
    "A conversation between User and Assistant. The user asks a question, and the Assistant solves it."
    "The assistant first thinks about the reasoning process in the mind and then provides the user with the answer."
    "The reasoning process and answer are enclosed within <think> </think> and <answer> </answer> tags, "
    "respectively, i.e., <think> reasoning process here </think> <answer> answer here </answer>. ")
TOOL_GUIDELINES = (
    "Integrate step-by-step reasoning and Python code to solve math problems using the following guidelines:\n\n"
    "- Analyze the question and write functions to solve the problem; the function should not take any arguments.\n"
    "- Present the final result in LaTeX using a `\\boxed{}` without any units.\n"
    "- Utilize the `pi` symbol and `Rational`` from Sympy for $\\pi$ and fractions, and simplify all fractions and "
    "square roots without converting them to decimal values.\n\n"
    "Here are some examples you may refer to:\n\n---\n\n")
ALPACA_HEADER = ("Below is an instruction that describes a task. Write a response that appropriately completes the "
                 "request.\n\n### Instruction:\n{instruction}\n\n### Response:\n")


# ------------------------------------------------------------------------------- chat wrappers
def chatml(user: str, system: Optional[str] = None, assistant: str = "") -> str:
    sys_turn = f"<|im_start|>system\n{system}<|im_end|>\n" if system else ""
    return f"{sys_turn}<|im_start|>user\n{user}<|im_end|>\n<|im_start|>assistant\n{assistant}"


def r1_turns(user: str, system: Optional[str] = None, assistant: str = "<think>\n") -> str:
    return f"<｜User｜>{user}<｜Assistant｜>{assistant}"


def tagged(user_tag: str, assistant_tag: str) -> Callable[..., str]:
    def wrap(user: str, system: Optional[str] = None, assistant: str = "") -> str:
        return f"{user_tag}{user}{assistant_tag}{assistant}"
    return wrap


@dataclass
class PromptStyle:
    """How one question (and optional worked examples) becomes the text the model continues."""
    name: str
    wrap: Callable[..., str]
    system: Optional[str] = None
    before: str = ""  # instruction in front of the question
    after: str = ""  # instruction behind the question
    assistant: str = ""  # what the assistant turn starts with
    separator: str = "\n\n"  # between worked examples
    answer_prefix: str = ""  # in front of a worked example's answer
    shots_in_one_turn: bool = False  # examples go INSIDE the user turn as plain text (chat models)
    preamble: str = ""  # once, in front of everything
    envelope: Optional[str] = None  # a template with {instruction} the finished prompt is placed in
    aliases: Tuple[str, ...] = field(default_factory=tuple)

    def question(self, q: str) -> str:
        return self.wrap(f"{self.before}{q}{self.after}", self.system, self.assistant)

    def example(self, q: str, a: str) -> str:
        return self.question(q) + self.answer_prefix + a

    def render(self, question: str, shots: Sequence[Sequence[str]] = ()) -> str:
        if shots and self.shots_in_one_turn:
            body = self.separator.join(f"{q}\n{a}" for q, a in shots) + self.separator + question
            text = self.question(body)
        elif shots:
            text = self.separator.join(self.example(q, a) for q, a in shots) + self.separator + self.question(question)
        else:
            text = self.question(question)
        if self.envelope:
            text = self.envelope.format(instruction=text)
        return (self.preamble + text).strip(" ")


def _styles() -> Dict[str, PromptStyle]:
    plain = tagged("", "")
    s: List[PromptStyle] = [
        PromptStyle("direct", tagged("Question: ", "\nAnswer: ")),
        PromptStyle("cot", tagged("Question: ", "\nAnswer: "), separator="\n\n\n"),
        PromptStyle("pal", tagged("Question: ", "\n\n"), separator="\n---\n"),
        PromptStyle("tool-integrated", tagged("Question: ", "\n\nSolution:\n"), separator="\n---\n",
                    preamble=TOOL_GUIDELINES),
        PromptStyle("self-instruct", tagged("<|user|>\n", "\n<|assistant|>\n"), separator="\n"),
        PromptStyle("tora", tagged("<|user|>\n", "\n<|assistant|>\n"), separator="\n", preamble=TOOL_GUIDELINES),
        PromptStyle("wizard_zs", tagged("### Instruction:\n", "\n\n### Response: Let's think step by step."),
                    separator="\n\n\n"),
        PromptStyle("platypus_fs", tagged("### Instruction:\n", "\n\n### Response:\n"), separator="\n\n\n",
                    envelope=ALPACA_HEADER),
        PromptStyle("deepseek-math", tagged("User: ", "\n\nAssistant:"), after="\n" + BOXED, separator="\n\n\n"),
        PromptStyle("kpmath", tagged("User: ", "\n\nAssistant:"),
                    before="Please reason step by step and " + ANSWER_IS + "\n\n", separator=""),
        PromptStyle("jiuzhang", tagged("## Question\n", "\n\n## Solution\n"), separator="\n\n\n"),
        PromptStyle("jiuzhang_tora", tagged("## Question\n", "\n\n## Code Solution\n"), separator="\n\n\n"),
        PromptStyle("jiuzhang_nl", tagged("## Question\n", "\n\n## Natural Language Solution\n"), separator="\n\n\n"),
        PromptStyle("mmiqc", tagged("", "\n\n"), before="Please solve the following problem and " + ANSWER_IS.replace(
            "final ", "") + "\n\n", separator="\n\n\n"),
        PromptStyle("abel", tagged("Question:\n", "\nAnswer:\nLet's think step by step.\n")),
        PromptStyle("shepherd", tagged("", "\n"), separator="\n\n\n"),
        PromptStyle("qwen-boxed", chatml, system="You are a helpful assistant.", after="\n" + BOXED),
        PromptStyle("qwen25-math-cot", chatml, system=BOXED, shots_in_one_turn=True),
        PromptStyle("qwen3-think", chatml, after="\n" + BOXED + "/think", assistant="<think>"),
        PromptStyle("qwen3", chatml, after="\n" + BOXED + "/no_think", assistant="<think></think>"),
        PromptStyle("qwen3-think-pure", chatml, after="\n/think", assistant="<think>"),
        PromptStyle("r1-distilled-qwen", r1_turns, after="\n" + BOXED, assistant="<think>\n",
                    aliases=("deepscaler-fix",)),
        PromptStyle("r1-distilled-pure", r1_turns, assistant="<think>\n"),
        PromptStyle("r1-distilled-qwen-gpqa", r1_turns, after="\n" + BOXED_CHOICE, assistant="<think>\n"),
        PromptStyle("r1-zero", tagged(R1_ZERO_PREAMBLE + "\nUser: ", "\nAssistant: \n")),
        PromptStyle("r1-zero-box", tagged(R1_ZERO_PREAMBLE + "\nUser: ", "\nAssistant: \n"),
                    after="\nPlease put your final answer within \\boxed{}."),
        PromptStyle("orz", tagged(R1_ZERO_PREAMBLE + "\nUser: ", "\nAssistant: <think>\n")),
        PromptStyle("qwq", chatml, system="You are a helpful and harmless assistant. You are Qwen developed by "
                                           "Alibaba. You should think step-by-step."),
        PromptStyle("qwq-math", chatml, system="You are a helpful and harmless assistant. You should think "
                                                "step-by-step, and put your final answer within \\boxed{}"),
        PromptStyle("nvidia-opencode", chatml,
                    system="You are Qwen, created by Alibaba Cloud. You are a helpful assistant.",
                    before="You are a helpful and harmless assistant. You should think step-by-step before responding "
                           "to the instruction below. Please use python programming language only.\n\nYou must use "
                           "```python for just the final solution code block.\n\n"),
        PromptStyle("mathstral", plain, after="\n" + BOXED),
        PromptStyle("internlm-math-fs", tagged("Question:", "\nAnswer:"), separator="\n"),
        PromptStyle("internlm-math-chat", chatml),
        PromptStyle("mistral", tagged("[INST] ", "[/INST]")),
        PromptStyle("numina", tagged("### Problem: ", "\n### Solution:"), answer_prefix=" "),
    ]
    table: Dict[str, PromptStyle] = {}
    for st in s:
        table[st.name] = st
        for a in st.aliases:
            table[a] = st
    return table


STYLES = _styles()

# benchmarks that share the worked examples of another one (reference: evaluation/utils.py:58-73)
SHOT_SOURCE = {**{k: "gsm8k" for k in ("gsm_hard", "svamp", "tabmwp", "asdiv", "mawps")},
               **{k: "math" for k in ("math_oai", "hungarian_exam", "math-oai", "aime24", "amc23")},
               "sat_math": "mmlu_stem",
               **{k: "gaokao" for k in ("gaokao2024_I", "gaokao2024_II", "gaokao_math_qa", "gaokao2024_mix",
                                        "cn_middle_school")}}


def style(name: str) -> PromptStyle:
    if name not in STYLES:
        raise KeyError(f"unknown prompt type {name!r}; known: {', '.join(sorted(STYLES))}")
    return STYLES[name]


def shots_for(data_name: str, examples: Optional[Dict[str, Sequence[Sequence[str]]]], num_shots: int
              ) -> Sequence[Sequence[str]]:
    """The first `num_shots` worked examples registered for the benchmark (or for the one it borrows from)."""
    if not num_shots or not examples:
        return ()
    return list(examples.get(SHOT_SOURCE.get(data_name, data_name), ()))[:num_shots]


def build_prompt(row: Dict, data_name: str, prompt_type: str, num_shots: int = 0,
                 examples: Optional[Dict[str, Sequence[Sequence[str]]]] = None, adapt_few_shot: bool = False) -> str:
    """Prompt text for one normalised benchmark row (``evaluation/benchmarks.py::normalise_row``).
    `adapt_few_shot`: Chinese exam sets always take five examples, and only their multiple-choice rows use them."""
    st = style(prompt_type)
    n = 5 if adapt_few_shot and SHOT_SOURCE.get(data_name) == "gaokao" else num_shots
    shots = shots_for(data_name, examples, n)
    if adapt_few_shot and str(row.get("gt_ans", row.get("answer", ""))) not in ("A", "B", "C", "D", "E"):
        shots = ()
    return st.render(row["question"], shots)
