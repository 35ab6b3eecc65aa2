// Configuration compiler, decoder topics only: what turns a user level Pheniqs job document into the compiled
// ontology the decoders are constructed from (the shape of reference test/BDGGG/valid/compile_annotated.out).
//
//   Transcode::apply_inheritance                 transcode.cpp:329-437   decoder repository, "base" references
//   Transcode::compile_transformation            transcode.cpp:735-763   default knit
//   Transcode::compile_decoder_transformation    transcode.cpp:824-940   cardinalities, random barcode probability,
//                                                                        undetermined, length / duplicate checks
//   Transcode::compile_decoder                   transcode.cpp:941-1039  index, concentration (1 - noise) / sum, BC, ID
//   CodecMetric::compile_barcode_tolerance       metric.h:87-111,216-242 shannon bound, default distance tolerance
//   projections "<type>:decoder"                 configuration.json:423-520
//
//   Job::load_instruction_with_import            job.cpp:167-224         `import`: documents merged depth first
//
// SAM header atoms and the command line stay with the caller.
#pragma once

#include "json.hpp"

namespace phq {

// true when every decoder of the document already went through the compiler
bool is_compiled_job(const Json& document);

// Transcode::assemble (transcode.cpp:323-349) without the file imports: the decoder repository and the `base`
// references resolved, the repository removed.  This is the "instruction" the prior estimator rewrites.
Json assemble_job(const Json& document);

// throws ConfigurationError exactly where the reference does
Json compile_job(const Json& document);

// Job::read_instruction_document (job.cpp:163-224): the document at `path` with its `import` list resolved — every
// imported document is loaded the same way (depth first, a path only the first time it is met), relative paths
// against the directory of the importing document; a later import takes precedence over an earlier one and the
// importing document over all of them (merge_json_value, json.cpp:780-803: what the base defines fills in what the
// document does not; a dictionary cannot fill a member that is not one).  The `import` member is removed.
Json load_job_document(const std::string& path);

// the same for a document that is already in memory; its imports are relative to `directory`
Json resolve_imports(Json document, const std::string& directory);

}  // namespace phq
