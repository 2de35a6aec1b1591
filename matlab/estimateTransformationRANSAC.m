function [model, inliers, isFound] = estimateTransformationRANSAC(matchedPoints1, matchedPoints2, transformType, input)
    %ESTIMATETRANSFORMATIONRANSAC Shadows PP/imageMatching/estimateTransformationRANSAC.m (all five transformTypes).
    %   The minimal-sample draws are generated here with randperm exactly as the reference does (:96) and handed to
    %   the device as an explicit input (a 4-row column per loop iteration, the first minPoints rows are the sample);
    %   fitting, scoring, the adaptive stop and the refit run in aps_mex.
    if nargin < 4, input = struct('maxDistance', 2.0, 'inliersConfidence', 99.9, 'maxIter', 500); end
    switch lower(transformType)  % getTransformParams (:648-660)
        case 'projective', k = 4;
        case 'affine', k = 3;
        case {'similarity', 'rigid'}, k = 2;
        case 'translation', k = 1;
        otherwise, error('Unknown transform type');
    end
    if size(matchedPoints1, 1) ~= size(matchedPoints2, 1)
        error('estimateTransformationRANSAC:PointCountMismatch', 'matchedPoints1 and matchedPoints2 must have the same number of rows.');
    end
    M = size(matchedPoints1, 1);
    if M < k
        model = []; inliers = false(M, 1); isFound = false; return;
    end
    S = input.maxIter + 64;
    sampleIdx = ones(4, S, 'uint32');
    for s = 1:S, sampleIdx(1:k, s) = uint32(randperm(M, k)); end
    [model, inliers, isFound] = aps_mex('ransac_homography', double(matchedPoints1), double(matchedPoints2), input, sampleIdx, lower(transformType));
    if ~isFound, model = []; end
end
