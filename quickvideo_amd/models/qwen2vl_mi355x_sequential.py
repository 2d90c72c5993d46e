"""Sequential variant (fetch all frames, then prefill) — counterpart of lvu/models/qwen25_lvu.py, for timing the
benefit of the overlap like the reference's timing_quickvideo.sh vs timing_quickvideo_interleaved.sh."""
from . import qwen2vl_mi355x as _m

init_lvu_model = _m.init_lvu_model


def chat_lvu_model(self, messages, **generation_kwargs):
    return _m.chat_lvu_model(self, messages, _overlap=False, **generation_kwargs)


def run_lvu_model(self, question, video_path, **generation_kwargs):
    return chat_lvu_model(self, _m.video_message(self.config, question, video_path), **generation_kwargs)
