"""ast_amd: MI355X-native (gfx950) implementation of the 0xSameer/ast encoder-decoder training hot path.
Layout: csrc/ (hand-written HIP kernels + the C ABI of include/astk.h), and the host-side mirror of the
reference's Python interface (seq2seq.SpeechEncoderDecoder, nn.NN, config.Config, dataloader, optimizers,
serializers).  See DESIGN.md."""
